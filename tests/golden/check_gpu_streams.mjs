// Runs every stream of tests/golden/gpu_streams.json (output of the GPU path, made by
// tools/make_gpu_streams.py on an MI355X) through the reference's own decoders under Node in
// the build container: RawInflate, Inflate{verify}, InflateStream, Gunzip of the per-module sources, Unzip
// {verify} of the bundle (the per-module unzip.js cannot read archives with several entries).
// The reference tree is read-only and lacks "type":"module": its src/ is copied to a scratch
// directory at run time, as tests/golden/gen_golden.mjs does.
//
//   node tests/golden/check_gpu_streams.mjs [/root/reference]
//
import fs from 'fs';
import os from 'os';
import path from 'path';
import crypto from 'crypto';
import { fileURLToPath, pathToFileURL } from 'url';

const REF = process.argv[2] || '/root/reference';
const here = path.dirname(fileURLToPath(import.meta.url));
const tmp = fs.mkdtempSync(path.join(os.tmpdir(), 'zref-'));
function cpdir(src, dst) {
	fs.mkdirSync(dst, { recursive: true });
	for (const e of fs.readdirSync(src, { withFileTypes: true })) {
		const s = path.join(src, e.name), d = path.join(dst, e.name);
		if (e.isDirectory()) cpdir(s, d);
		else fs.copyFileSync(s, d);
	}
}
cpdir(path.join(REF, 'src'), path.join(tmp, 'src'));
fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
const imp = (p) => import(pathToFileURL(path.join(tmp, p)).href);
const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8)).digest('hex');

(async () => {
	const { RawInflate } = await imp('src/rawinflate.js');
	const { Inflate } = await imp('src/inflate.js');
	const { Gunzip } = await imp('src/gunzip.js');
	const { InflateStream } = await imp('src/inflate_stream.js');
	const bundle = await imp('src/zlib.es.js');
	const Unzip = bundle.Unzip || (bundle.Zlib && bundle.Zlib.Unzip) || (bundle.default && bundle.default.Unzip);
	const fix = JSON.parse(fs.readFileSync(path.join(here, 'gpu_streams.json'), 'utf8'));
	let ok = 0, streamed = 0;
	const bad = [];
	for (const s of fix.streams) {
		const comp = new Uint8Array(Buffer.from(s.b64, 'base64'));
		const tag = s.kind + ' ' + JSON.stringify(s.spec || s.entries.map((e) => e.name)) + (s.kind === 'raw' ? ' ctype ' + s.ctype + ' lazy ' + s.lazy : '') + (s.chunked ? ' chunked' : '');
		try {
			if (s.kind === 'raw') {
				// a bare raw stream needs trailing pad bytes (the decoder reads ahead)
				const padded = new Uint8Array(comp.length + 8);
				padded.set(comp);
				const out = new RawInflate(padded, { bufferSize: Math.max(s.n, 1) }).decompress();
				if (out.length !== s.n || sha(out) !== s.sha256) throw new Error('content differs: ' + out.length + ' bytes');
			} else if (s.kind === 'zlib') {
				const out = new Inflate(comp, { verify: true, bufferSize: Math.max(s.n, 1) }).decompress();
				if (out.length !== s.n || sha(out) !== s.sha256) throw new Error('content differs: ' + out.length + ' bytes');
				// and through the streaming inflater, fed in small pieces as the reference's own
				// "(stream)" cases feed it (test/browser.es/staff/base.js:32-52)
				const st = new InflateStream();
				const parts = [];
				let total = 0;
				for (let i = 0; i < comp.length; i += 997) {
					const buf = st.decompress(comp.subarray(i, Math.min(comp.length, i + 997)));
					parts.push(new Uint8Array(buf)); // the inflater reuses its buffer: keep a copy
					total += buf.length;
				}
				const whole = new Uint8Array(total);
				let at = 0;
				for (const p of parts) {
					whole.set(p, at);
					at += p.length;
				}
				if (total !== s.n || sha(whole) !== s.sha256) throw new Error('InflateStream: content differs: ' + total + ' bytes');
				++streamed;
			} else if (s.kind === 'gzip') {
				const out = new Gunzip(comp).decompress(); // checks CRC-32 and ISIZE
				if (out.length !== s.n || sha(out) !== s.sha256) throw new Error('content differs: ' + out.length + ' bytes');
			} else if (s.kind === 'zip') {
				// an archive made under a password is decrypted by the reference's own Unzip (src/unzip.js:112-119)
				const uz = new Unzip(comp, s.password ? { verify: true, password: new Uint8Array(s.password) } : { verify: true });
				const names = uz.getFilenames();
				if (names.length !== s.entries.length) throw new Error('entries: ' + names.join(','));
				for (const e of s.entries) {
					const out = uz.decompress(e.name);
					if (out.length !== e.n || sha(out) !== e.sha256) throw new Error('entry ' + e.name + ' differs');
				}
			} else throw new Error('kind ' + s.kind);
			++ok;
		} catch (e) {
			bad.push(tag + ': ' + (e && e.message ? e.message : e));
		}
	}
	console.log(JSON.stringify({ streams: fix.streams.length, ok, streamed, bad }));
	process.exit(bad.length ? 1 : 0);
})().catch((e) => {
	console.error(e);
	process.exit(2);
});
