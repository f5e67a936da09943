// Generates tests/golden/golden.json by running the reference implementation
// (per-module ES sources of ryunosinfx/zlib.es) under Node in the build
// container.  The reference tree is read-only and lacks "type":"module", so
// its src/ and vendor/mt.js are copied to a scratch directory at run time;
// nothing from it is written into this repository except the input specs and
// the outputs it produced (hex / sha256 / sizes).
//
//   node tests/golden/gen_golden.mjs [/root/reference]
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
cpdir(path.join(REF, 'vendor', 'mt.js'), path.join(tmp, 'vendor', 'mt.js'));
fs.writeFileSync(path.join(tmp, 'package.json'), '{"type":"module"}');
const imp = (p) => import(pathToFileURL(path.join(tmp, p)).href);
// Node 12 has no top-level await
(async () => {
const { RawDeflate } = await imp('src/rawdeflate.js');
const { Deflate } = await imp('src/deflate.js');
const { Gzip } = await imp('src/gzip.js');
const { Zip } = await imp('src/zip.js');
const { CRC32 } = await imp('src/crc32.js');
const { Adler32 } = await imp('src/adler32.js');
const { MersenneTwister } = await imp('vendor/mt.js/mt.es.js');

const hex = (u8) => Buffer.from(u8).toString('hex');
const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8)).digest('hex');

// ---- input generators (twins of oracle/zo_gen.c) ----
function genRnd(n, seed) {
	const mt = new MersenneTwister(seed), d = new Uint8Array(n);
	for (let i = 0; i < n; ++i) d[i] = mt.nextInt(256);
	return d;
}
function genSeq(n) {
	const d = new Uint8Array(n);
	for (let i = 0; i < n; ++i) d[i] = i & 0xff;
	return d;
}
function genRndSeq(n, seed) {
	const mt = new MersenneTwister(seed), d = new Uint8Array(n);
	for (let i = 0; i < n; ) {
		let r1 = mt.nextInt(256), r2 = mt.nextInt(256);
		while (r2--) {
			if (i === n) break;
			d[i++] = r1++ & 0xff;
		}
	}
	return d;
}
function genText(n, seed) {
	const lcg = (s) => (Math.imul(s, 1664525) + 1013904223) >>> 0;
	const vocab = [], d = new Uint8Array(n);
	let v = 0x9e3779b9;
	for (let w = 0; w < 4096; ++w) {
		v = lcg(v);
		const len = 2 + ((v >>> 16) % 9), word = [];
		for (let k = 0; k < len; ++k) {
			v = lcg(v);
			word.push(97 + ((v >>> 16) % 26));
		}
		vocab.push(word);
	}
	let x = (seed ^ 0x5bd1e995) >>> 0, i = 0, wc = 0;
	while (i < n) {
		x = lcg(x);
		const r = x >>> 20, rank = Math.floor((r * r * r) / 16777216), word = vocab[rank];
		for (let k = 0; k < word.length && i < n; ++k) d[i++] = word[k];
		if (i < n) d[i++] = ++wc % 12 === 0 ? 10 : 32;
	}
	return d;
}
function genRuns(n, seed) {
	// low-entropy: long runs of few byte values
	const mt = new MersenneTwister(seed), d = new Uint8Array(n);
	for (let i = 0; i < n; ) {
		const b = mt.nextInt(4) * 85, run = 1 + mt.nextInt(512);
		for (let k = 0; k < run && i < n; ++k) d[i++] = b;
	}
	return d;
}
function make(spec) {
	switch (spec.gen) {
		case 'bytes': return new Uint8Array(Buffer.from(spec.hex, 'hex'));
		case 'rnd': return genRnd(spec.n, spec.seed);
		case 'seq': return genSeq(spec.n);
		case 'rndseq': return genRndSeq(spec.n, spec.seed);
		case 'text': return genText(spec.n, spec.seed);
		case 'runs': return genRuns(spec.n, spec.seed);
		case 'small-alpha': {
			const mt = new MersenneTwister(spec.seed), d = new Uint8Array(spec.n);
			for (let i = 0; i < spec.n; ++i) d[i] = 97 + mt.nextInt(spec.k);
			return d;
		}
	}
	throw new Error('gen ' + spec.gen);
}

function rawdef(input, ctype, lazy) {
	// fresh copy: RawDeflate keeps a reference to its input
	const o = { compressionType: ctype };
	if (lazy) o.lazy = lazy;
	return new RawDeflate(input, o).compress();
}
function record(u8, full) {
	const r = { len: u8.length, sha256: sha(u8) };
	if (full || u8.length <= 96) r.hex = hex(u8);
	return r;
}

const out = { generator: 'tests/golden/gen_golden.mjs', reference: 'ryunosinfx/zlib.es src/*.js (per-module)', cases: [] };

// ---- 1. named cases (reference test inputs and sizes) ----
const specs = [
	{ gen: 'bytes', hex: '' },
	{ gen: 'bytes', hex: '41' },
	{ gen: 'bytes', hex: '4141' },
	{ gen: 'bytes', hex: '414141' },
	{ gen: 'bytes', hex: '41414141' },
	{ gen: 'bytes', hex: hex(Uint8Array.from({ length: 23 }, (_, i) => i + 1)) }, // zlib-test.js "undercomitted"
	{ gen: 'bytes', hex: hex(Buffer.from('abcabcabcabcXabcabc')) },
	{ gen: 'bytes', hex: hex(Buffer.from('123456789')) },
	{ gen: 'bytes', hex: hex(Buffer.from('hoge\n')) },
	{ gen: 'bytes', hex: hex(Buffer.from('Wikipedia')) },
	{ gen: 'bytes', hex: hex(Buffer.from('hello world\n')) },
	{ gen: 'seq', n: 76543 },
	{ gen: 'rnd', n: 76543, seed: 1 },
	{ gen: 'rndseq', n: 76543, seed: 1346432776267 },
	{ gen: 'rndseq', n: 65536, seed: 1 },
	{ gen: 'seq', n: 123456 },
	{ gen: 'text', n: 65536, seed: 42 },
	{ gen: 'text', n: 16384, seed: 7 },
	{ gen: 'text', n: 262144, seed: 42 },
	{ gen: 'runs', n: 65536, seed: 3 },
	{ gen: 'bytes', hex: '00'.repeat(1000) },
	{ gen: 'small-alpha', n: 20000, seed: 5, k: 2 },
	{ gen: 'small-alpha', n: 70000, seed: 6, k: 3 },
];
for (const spec of specs) {
	const input = make(spec);
	const c = { spec, n: input.length, crc32: CRC32.calc(input), adler32: Adler32.mkHash(input), raw: {} };
	for (const ct of [0, 1, 2]) c.raw[ct] = record(rawdef(input, ct, 0));
	if (input.length <= 64) {
		const rd = new RawDeflate(input, { compressionType: 2 });
		c.tokens = Array.from(rd.lz77(input));
	}
	c.zlib = {};
	for (const ct of [0, 1, 2]) c.zlib[ct] = record(new Deflate(input, { compressionType: ct }).compress());
	out.cases.push(c);
}

// ---- 2. lazy variants (including the reference's tail behaviour) ----
out.lazy = [];
for (const spec of [
	{ gen: 'text', n: 4096, seed: 11 },
	{ gen: 'text', n: 119, seed: 42 },
	{ gen: 'text', n: 30000, seed: 12 },
	{ gen: 'rndseq', n: 20000, seed: 9 },
	{ gen: 'small-alpha', n: 5000, seed: 2, k: 4 },
])
	for (const lazy of [4, 8, 32, 10000])
		for (const ct of [1, 2]) {
			const input = make(spec);
			out.lazy.push({ spec, lazy, ctype: ct, out: record(rawdef(input, ct, lazy)) });
		}

// ---- 3. fuzz: many short inputs, sha256 only ----
out.fuzz = [];
{
	const mt = new MersenneTwister(20250301);
	for (let k = 0; k < 400; ++k) {
		const kind = ['rnd', 'rndseq', 'text', 'runs', 'small-alpha'][mt.nextInt(5)];
		const n = k < 40 ? k : 1 + mt.nextInt(k % 7 === 0 ? 9000 : 700);
		const spec = { gen: kind, n, seed: mt.nextInt(1 << 30) };
		if (kind === 'small-alpha') spec.k = 1 + mt.nextInt(6);
		const input = make(spec);
		const lazy = k % 5 === 4 ? [0, 4, 16][mt.nextInt(3)] : 0;
		const r = { spec, lazy, crc32: CRC32.calc(input), adler32: Adler32.mkHash(input) };
		for (const ct of [0, 1, 2]) {
			const o = rawdef(input, ct, lazy);
			r['l' + ct] = o.length;
			r['s' + ct] = sha(o).slice(0, 16);
		}
		out.fuzz.push(r);
	}
}

// ---- 4. getLengths_ / getCodesFromLengths_ known answers ----
out.lengths = [];
{
	const rd = new RawDeflate(new Uint8Array(0));
	const mt = new MersenneTwister(99);
	const tables = [];
	for (let k = 0; k < 60; ++k) {
		const nsym = [286, 30, 19][k % 3], limit = [15, 7, 7][k % 3];
		const f = new Uint32Array(nsym);
		const used = 1 + mt.nextInt(nsym);
		for (let j = 0; j < used; ++j) {
			const shape = k % 4;
			const v = shape === 0 ? 1 + mt.nextInt(8) : shape === 1 ? 1 + mt.nextInt(1000) : shape === 2 ? 1 << mt.nextInt(17) : 60000 + mt.nextInt(12000);
			f[mt.nextInt(nsym)] = v;
		}
		tables.push({ f, limit });
	}
	// Fibonacci-like frequencies force the length limit
	const fib = new Uint32Array(286);
	for (let i = 0, a = 1, b = 1; i < 40; ++i) { fib[i] = a; [a, b] = [b, (a + b) % 60000]; }
	tables.push({ f: fib, limit: 15 });
	const fib2 = new Uint32Array(30);
	for (let i = 0, a = 1, b = 1; i < 20; ++i) { fib2[i] = a; [a, b] = [b, a + b]; }
	tables.push({ f: fib2, limit: 7 });
	// exact multiples of 65536 vanish in the Uint16 heap
	const wrap = new Uint32Array(286);
	wrap[10] = 65536; wrap[11] = 65537; wrap[12] = 70000; wrap[13] = 5; wrap[256] = 2;
	tables.push({ f: wrap, limit: 15 });
	for (const t of tables) {
		const l = rd.getLengths_(t.f, t.limit);
		const c = rd.getCodesFromLengths_(l);
		out.lengths.push({ freqs: Array.from(t.f), limit: t.limit, lengths: Array.from(l), codes: Array.from(c) });
	}
}

// ---- 5. gzip (Date.now pinned) ----
out.gzip = [];
{
	const realNow = Date.now;
	Date.now = () => 1346432776267;
	for (const [spec, opt] of [
		[{ gen: 'bytes', hex: hex(Buffer.from('aaaa')) }, {}],
		[{ gen: 'bytes', hex: hex(Buffer.from('hello world\n')) }, { deflateOptions: { compressionType: 1 } }],
		[{ gen: 'rndseq', n: 76543, seed: 1346432776267 }, { flags: { fname: true, fhcrc: true }, filename: 'foobar.filename' }],
		[{ gen: 'text', n: 5000, seed: 3 }, { flags: { fname: true, fcomment: true, comment: true }, filename: 'a.txt', comment: 'note', deflateOptions: { compressionType: 2 } }],
		[{ gen: 'bytes', hex: '' }, { deflateOptions: { compressionType: 0 } }],
	]) {
		const o = new Gzip(make(spec), JSON.parse(JSON.stringify(opt))).compress();
		out.gzip.push({ spec, opt, mtime: (1346432776267 / 1000) | 0, out: record(o, o.length <= 4096) });
	}
	Date.now = realNow;
}

// ---- 6. zip framing (fixed date; STORE and DEFLATE entries) ----
out.zip = [];
{
	const date = new Date(2013, 4, 17, 10, 31, 44);
	const enc = (s) => new Uint8Array(Buffer.from(s));
	const files = [
		{ name: 'hoge.txt', spec: { gen: 'bytes', hex: hex(Buffer.from('hoge\n')) }, method: 0 },
		{ name: 'dir/text.txt', spec: { gen: 'text', n: 3000, seed: 5 }, method: 8, ctype: 2 },
		{ name: 'seq.bin', spec: { gen: 'seq', n: 1000 }, method: 8, ctype: 1, comment: 'c' },
	];
	const zip = new Zip({ comment: enc('archive') });
	for (const f of files) {
		const o = { filename: enc(f.name), compressionMethod: f.method, date };
		if (f.comment) o.comment = enc(f.comment);
		if (f.ctype !== undefined) o.deflateOption = { compressionType: f.ctype };
		zip.addFile(make(f.spec), o);
	}
	const o = zip.compress();
	out.zip.push({ files, date: [2013, 5, 17, 10, 31, 44], comment: 'archive', out: record(o, true) });
}

// ---- 7. ZipCrypto (src/zip.js:138-147,287-318): the cipher alone, and an archive of STORE entries under a
// password with Math.random pinned (thirteen files: the header loop tests the file index against 11) ----
out.zipcrypto = [];
{
	const z = new Zip();
	for (const [pw, spec] of [
		[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9], { gen: 'seq', n: 300 }],
		[Array.from(Buffer.from('hogefugapiyo')), { gen: 'rnd', n: 1000, seed: 7 }],
	]) {
		const key = z.createEncryptionKey(pw);
		const keyAfterPassword = Array.from(key);
		const plain = make(spec);
		const enc = new Uint8Array(plain.length);
		for (let i = 0; i < plain.length; ++i) enc[i] = z.encode(key, plain[i]);
		out.zipcrypto.push({ password: pw, spec, key_after_password: keyAfterPassword, key_after: Array.from(key), out: record(enc, true) });
	}
	const realRandom = Math.random;
	let st = 12345;
	Math.random = () => {
		st = (st * 48271) % 2147483647;
		return st / 2147483647;
	};
	const date = new Date(2013, 4, 17, 10, 31, 44);
	const enc = (s) => new Uint8Array(Buffer.from(s));
	const zip = new Zip();
	const files = [];
	for (let i = 0; i < 13; ++i) {
		const spec = { gen: 'rnd', n: 20 + 7 * i, seed: 100 + i };
		const o = { filename: enc('f' + i + '.bin'), compressionMethod: 0, date };
		if (i === 5) o.password = [9, 8, 7];
		zip.addFile(make(spec), o);
		files.push({ name: 'f' + i + '.bin', spec, password: i === 5 ? [9, 8, 7] : null });
	}
	zip.setPassword([0, 1, 2, 3, 4, 5, 6, 7, 8, 9]);
	const o = zip.compress();
	Math.random = realRandom;
	out.zipcrypto_archive = { files, password: [0, 1, 2, 3, 4, 5, 6, 7, 8, 9], date: [2013, 5, 17, 10, 31, 44], random: 'st = st * 48271 % 2147483647 from 12345; st / 2147483647', out: record(o, true) };
}

fs.writeFileSync(path.join(here, 'golden.json'), JSON.stringify(out));
fs.rmSync ? fs.rmSync(tmp, { recursive: true, force: true }) : fs.rmdirSync(tmp, { recursive: true });
console.log('wrote golden.json:', out.cases.length, 'cases,', out.lazy.length, 'lazy,', out.fuzz.length, 'fuzz,', out.lengths.length, 'length tables');
})().catch((e) => { console.error(e); process.exit(1); });
