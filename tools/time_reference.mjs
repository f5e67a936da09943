// Times the reference's own RawDeflate (per-module sources under Node, one core) on an input file:
// the whole file as one call, and the same bytes as independent 64 KiB units.  Build container only
// (the reference tree is not on the GPU box).  The tree is read-only and lacks "type":"module": its
// src/ is copied to a scratch directory at run time.
//   node tools/time_reference.mjs <input file> [/root/reference]
import fs from 'fs';
import os from 'os';
import path from 'path';
import { pathToFileURL } from 'url';

const input = new Uint8Array(fs.readFileSync(process.argv[2]));
const REF = process.argv[3] || '/root/reference';
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
(async () => {
	const { RawDeflate } = await import(pathToFileURL(path.join(tmp, 'src/rawdeflate.js')).href);
	const best = (fn) => {
		let b = Infinity, r;
		for (let k = 0; k < 4; ++k) {
			const t = process.hrtime.bigint();
			r = fn();
			b = Math.min(b, Number(process.hrtime.bigint() - t) / 1e9);
		}
		return [b, r];
	};
	const res = { node: process.version, input_bytes: input.length };
	for (const ct of [1, 2]) {
		const [tw, whole] = best(() => new RawDeflate(input, { compressionType: ct }).compress().length);
		const [tu, units] = best(() => {
			let n = 0;
			for (let o = 0; o < input.length; o += 65536) n += new RawDeflate(input.subarray(o, Math.min(o + 65536, input.length)), { compressionType: ct }).compress().length;
			return n;
		});
		res['ctype' + ct] = { whole_s: +tw.toFixed(4), whole_MBps: +(input.length / tw / 1e6).toFixed(2), whole_out: whole,
			units64k_s: +tu.toFixed(4), units64k_MBps: +(input.length / tu / 1e6).toFixed(2), units64k_out: units };
	}
	console.log(JSON.stringify(res));
})();
