'use strict';
// The reference's compression tests restated against the GPU-backed shim
// (cases of test/browser.es/zlib-test.js, raw-inflate-test.js,
// code-path-test.js, gzip-test.js, zip-test.js and test/nodejs/node-test.js).
// Decoding is done by Node's built-in zlib, as test/nodejs/node-test.js:195-221
// does for cross-implementation conformance.  Needs a GPU.
const assert = require('assert');
const nodeZlib = require('zlib');
const { Zlib, deflateChunked, gzipChunked, zlibChunked } = require('../zlib_gpu.js');
const H = require('./helpers.js');
const G = require('../zlib_gpu.js');

const size = 76543; // test/browser.es/zlib-test.js:7
const CT = Zlib.RawDeflate.CompressionType;
let passed = 0;
function it(name, fn) {
	fn();
	++passed;
	console.log('ok  ' + name);
}
const inputs = {
	random: H.makeRandomData(size, 20121006),
	sequential: H.makeSequentialData(size),
	'random sequential': H.makeRandomSequentialData(size, 1346432776267),
};

// ---- checksums (src/crc32.js, src/adler32.js) ----
it('CRC32 / Adler32 known answers', () => {
	assert.strictEqual(Zlib.CRC32.calc(Buffer.from('123456789')), 0xcbf43926);
	assert.strictEqual(Zlib.CRC32.calc(Buffer.from('hoge\n')), 0xb9e2879d); // unzip-test.js stored entry
	assert.strictEqual(Zlib.Adler32.mkHash('Wikipedia'), 0x11e60398);
	for (const k in inputs) {
		assert.strictEqual(Zlib.CRC32.calc(inputs[k]), H.crc32(inputs[k]), k);
		assert.strictEqual(Zlib.Adler32.mkHash(inputs[k]), H.adler32(inputs[k]), k);
	}
	const d = inputs.random;
	assert.strictEqual(Zlib.CRC32.update(d.subarray(1000), Zlib.CRC32.calc(d.subarray(0, 1000))), H.crc32(d));
	assert.strictEqual(Zlib.Adler32.update(Zlib.Adler32.mkHash(d.subarray(0, 1000)), d.subarray(1000)), H.adler32(d));
	assert.strictEqual(Zlib.CRC32.calc(d, 10, 500), H.crc32(d.subarray(10, 510)));
	// a position without a length runs to the end of the array (src/crc32.js:25-29)
	assert.strictEqual(Zlib.CRC32.update(d, Zlib.CRC32.calc(d.subarray(0, 777)), 777), H.crc32(d));
	assert.strictEqual(Zlib.RawDeflate.CompressionType.UNCOMPRESSED, Zlib.RawDeflate.CompressionType.NONE);
});

// ---- zlib-test.js:15-84 / raw-inflate-test.js:25-108 ----
for (const tname of ['NONE', 'FIXED', 'DYNAMIC'])
	for (const k in inputs) {
		it(`Deflate -> inflate: ${k} data, ${tname}`, () => {
			const deflated = new Zlib.Deflate(inputs[k], { compressionType: CT[tname] }).compress();
			const inflated = nodeZlib.inflateSync(Buffer.from(deflated)); // verifies the Adler-32 trailer
			assert.ok(H.equal(inflated, inputs[k]));
			assert.strictEqual(deflated.byteOffset, 0);
		});
		it(`RawDeflate -> rawinflate: ${k} data, ${tname}`, () => {
			const deflated = new Zlib.RawDeflate(inputs[k], { compressionType: CT[tname] }).compress();
			assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.from(deflated)), inputs[k]));
		});
	}
it('undercomitted', () => {
	const data = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23];
	const compressed = new Zlib.Deflate(data).compress();
	assert.ok(H.equal(nodeZlib.inflateSync(Buffer.from(compressed)), data));
	assert.ok(H.equal(nodeZlib.inflateSync(Buffer.from(Zlib.Deflate.compress(data))), data));
});

// ---- code-path-test.js:21-31,41-202: the requested block maker is the one that runs ----
it('code path: block makers are the executed call sites', () => {
	const names = ['makeNocompressBlock', 'makeFixedHuffmanBlock', 'makeDynamicHuffmanBlock'];
	const calls = {};
	const orig = {};
	for (const nm of names) {
		orig[nm] = Zlib.RawDeflate.prototype[nm];
		calls[nm] = 0;
		Zlib.RawDeflate.prototype[nm] = function () {
			calls[nm]++;
			return orig[nm].apply(this, arguments);
		};
	}
	try {
		const want = { NONE: names[0], FIXED: names[1], DYNAMIC: names[2] };
		for (const tname in want) {
			for (const nm of names) calls[nm] = 0;
			const out = new Zlib.Deflate(inputs.sequential, { compressionType: CT[tname] }).compress();
			for (const nm of names) assert.strictEqual(calls[nm] > 0, nm === want[tname], `${tname}/${nm}`);
			assert.ok(H.equal(nodeZlib.inflateSync(Buffer.from(out)), inputs.sequential));
		}
	} finally {
		for (const nm of names) Zlib.RawDeflate.prototype[nm] = orig[nm];
	}
	assert.throws(() => new Zlib.RawDeflate(inputs.sequential, { compressionType: 3 }).compress());
});

// ---- lazy option (rawdeflate.js:86-87,339-347) ----
it('lazy matching round trips and does not grow the output', () => {
	const text = H.makeText(262144, 5);
	const s0 = new Zlib.RawDeflate(text, { lazy: 0 }).compress();
	const s32 = new Zlib.RawDeflate(text, { lazy: 32 }).compress();
	assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.from(s32)), text));
	assert.ok(s32.length <= s0.length);
	const ref = nodeZlib.deflateRawSync(Buffer.from(text), { level: 6 });
	assert.ok(s0.length < ref.length * 1.1, `${s0.length} vs zlib-6 ${ref.length}`);
});

// ---- gzip-test.js:21-76 ----
it('Gzip -> gunzip (random sequential, fixed seed)', () => {
	const data = inputs['random sequential'];
	assert.ok(H.equal(nodeZlib.gunzipSync(Buffer.from(new Zlib.Gzip(data).compress())), data));
});
it('Gzip with filename, comment and header crc', () => {
	const data = inputs.sequential;
	const out = new Zlib.Gzip(data, { flags: { fname: true, fcomment: true, comment: true, fhcrc: true }, filename: 'foobar.filename', comment: 'hello' }).compress();
	assert.strictEqual(out[3], 0x08 | 0x10 | 0x02);
	assert.strictEqual(Buffer.from(out.subarray(10, 25)).toString(), 'foobar.filename');
	assert.ok(H.equal(nodeZlib.gunzipSync(Buffer.from(out)), data));
});

// ---- zip-test.js:17-152 (STORE / DEFLATE entries; central directory parsed here) ----
// traditional PKZIP decryption (the stream cipher of src/unzip.js:113-117, written out for the test):
// the twelve header bytes are dropped unchecked, as the reference's Unzip drops them
function zipDecrypt(body, password) {
	const key = [305419896, 591751049, 878082192];
	const crcTab = new Uint32Array(256);
	for (let i = 0; i < 256; ++i) {
		let c = i;
		for (let k = 0; k < 8; ++k) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1;
		crcTab[i] = c >>> 0;
	}
	const upd = (n) => {
		key[0] = (crcTab[(key[0] ^ n) & 0xff] ^ (key[0] >>> 8)) >>> 0;
		key[1] = (Math.imul(key[1] + (key[0] & 0xff), 134775813) + 1) >>> 0;
		key[2] = (crcTab[(key[2] ^ (key[1] >>> 24)) & 0xff] ^ (key[2] >>> 8)) >>> 0;
	};
	for (const b of password) upd(b & 0xff);
	const out = new Uint8Array(body.length);
	for (let i = 0; i < body.length; ++i) {
		const t = (key[2] & 0xffff) | 2;
		out[i] = body[i] ^ ((Math.imul(t, t ^ 1) >>> 8) & 0xff);
		upd(out[i]);
	}
	return out.subarray(12);
}
function unzip(z, passwords) {
	let p = z.length - 22;
	while (p >= 0 && !(z[p] === 0x50 && z[p + 1] === 0x4b && z[p + 2] === 5 && z[p + 3] === 6)) --p;
	assert.ok(p >= 0, 'end of central directory');
	const dv = new DataView(z.buffer, z.byteOffset, z.byteLength);
	const count = dv.getUint16(p + 10, true);
	let cd = dv.getUint32(p + 16, true);
	const files = {};
	for (let i = 0; i < count; ++i) {
		assert.strictEqual(dv.getUint32(cd, true), 0x02014b50);
		const method = dv.getUint16(cd + 10, true), crc = dv.getUint32(cd + 16, true);
		const csize = dv.getUint32(cd + 20, true), usize = dv.getUint32(cd + 24, true);
		const nlen = dv.getUint16(cd + 28, true), xlen = dv.getUint16(cd + 30, true), clen = dv.getUint16(cd + 32, true);
		const off = dv.getUint32(cd + 42, true);
		const name = Buffer.from(z.subarray(cd + 46, cd + 46 + nlen)).toString();
		assert.strictEqual(dv.getUint32(off, true), 0x04034b50);
		const lnlen = dv.getUint16(off + 26, true), lxlen = dv.getUint16(off + 28, true);
		let body = z.subarray(off + 30 + lnlen + lxlen, off + 30 + lnlen + lxlen + csize);
		if (passwords) {
			assert.strictEqual(dv.getUint16(off + 6, true) & 1, 1, 'ENCRYPT flag, local header');
			assert.strictEqual(dv.getUint16(cd + 8, true) & 1, 1, 'ENCRYPT flag, central directory');
			body = zipDecrypt(body, passwords[name] || passwords['*']);
		}
		const plain = method === 8 ? nodeZlib.inflateRawSync(Buffer.from(body)) : body;
		assert.strictEqual(plain.length, usize, name);
		assert.strictEqual(H.crc32(plain), crc, name);
		files[name] = plain;
		cd += 46 + nlen + xlen + clen;
	}
	return files;
}
it('Zip: STORE and DEFLATE entries, batched compress()', () => {
	const td = { hogehoge: H.makeText(30000, 1), fugafuga: inputs.sequential.subarray(0, 5000), piyopiyo: inputs.random.subarray(0, 4000), empty: new Uint8Array(0) };
	const zip = new Zlib.Zip();
	let i = 0;
	for (const key in td) {
		const opt = { filename: new Uint8Array(Buffer.from(key)) };
		if (i % 3 === 1) opt.compressionMethod = Zlib.Zip.CompressionMethod.STORE;
		if (i % 3 === 2) opt.deflateOption = { compressionType: CT.FIXED };
		zip.addFile(td[key], opt);
		++i;
	}
	const files = unzip(zip.compress());
	for (const key in td) assert.ok(H.equal(files[key], td[key]), key);
});
// zip-test.js:83-150 (compress with password: one for the archive, then one per file)
it('Zip: compress with password (deflate)', () => {
	const td = { hogehoge: H.makeText(30000, 1), fugafuga: inputs.sequential.subarray(0, 5000), piyopiyo: inputs.random.subarray(0, 4000) };
	const zip = new Zlib.Zip();
	zip.setPassword([0, 1, 2, 3, 4, 5, 6, 7, 8, 9]);
	for (const key in td) zip.addFile(td[key], { filename: new Uint8Array(Buffer.from(key)) });
	const files = unzip(zip.compress(), { '*': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9] });
	for (const key in td) assert.ok(H.equal(files[key], td[key]), key);
});
it('Zip: compress with password (each file)', () => {
	const td = { hogehoge: [H.makeText(30000, 1), [1, 2, 3]], fugafuga: [inputs.sequential.subarray(0, 5000), [4, 5]], piyopiyo: [inputs.random.subarray(0, 4000), [6]] };
	const zip = new Zlib.Zip();
	const pw = {};
	for (const key in td) {
		zip.addFile(td[key][0], { filename: new Uint8Array(Buffer.from(key)), password: td[key][1], deflateOption: { compressionType: CT.FIXED } });
		pw[key] = td[key][1];
	}
	const files = unzip(zip.compress(), pw);
	for (const key in td) assert.ok(H.equal(files[key], td[key][0]), key);
});
it('Zip: many small files in one batch', () => {
	const zip = new Zlib.Zip();
	const td = [];
	const mt = new H.MT(99);
	for (let f = 0; f < 300; ++f) {
		const n = 1 + (mt.next() % 16384);
		const d = f % 4 === 3 ? H.makeRandomData(n, f) : H.makeText(n, f);
		td.push(d);
		zip.addFile(d, { filename: new Uint8Array(Buffer.from('f' + f)), deflateOption: { compressionType: f & 1 ? CT.FIXED : CT.DYNAMIC } });
	}
	const files = unzip(zip.compress());
	td.forEach((d, f) => assert.ok(H.equal(files['f' + f], d), 'f' + f));
});

// ---- chunked single-member streams (sync-aligned chunks + checksum combine) ----
it('gzipChunked: one member, many chunks compressed side by side', () => {
	const data = H.makeText(3 * 1048576 + 12345, 7);
	const gz = gzipChunked(data, { chunkSize: 1 << 20, mtime: 0 });
	assert.ok(H.equal(nodeZlib.gunzipSync(Buffer.from(gz)), data)); // checks CRC-32 and ISIZE
	const d = deflateChunked(data, { chunkSize: 65536 });
	assert.strictEqual(d.crc32, H.crc32(data));
	assert.strictEqual(d.adler32, H.adler32(data));
	assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.concat(d.parts.map((p) => Buffer.from(p)))), data));
	// independent chunks: every chunk but the last decodes alone once given a final empty block
	const ind = deflateChunked(data, { chunkSize: 65536, independent: true });
	const alone = Buffer.concat([Buffer.from(ind.parts[1]), Buffer.from([1, 0, 0, 0xff, 0xff])]);
	assert.ok(H.equal(nodeZlib.inflateRawSync(alone), data.subarray(65536, 131072)));
	const size = (r) => r.parts.reduce((s, p) => s + p.length, 0);
	assert.ok(size(d) <= size(ind));
});

it('zlibChunked: one zlib stream, chunks compressed side by side, Adler-32 by combine', () => {
	const data = H.makeText(2 * 1048576 + 777, 9);
	for (const ct of [CT.FIXED, CT.DYNAMIC]) {
		const zs = zlibChunked(data, { chunkSize: 262144, compressionType: ct });
		assert.strictEqual(zs[0], 0x78);
		assert.strictEqual(((zs[0] << 8) | zs[1]) % 31, 0);
		assert.ok(H.equal(nodeZlib.inflateSync(Buffer.from(zs)), data)); // verifies the Adler-32 trailer
	}
	assert.ok(H.equal(nodeZlib.inflateSync(Buffer.from(zlibChunked(new Uint8Array(0)))), new Uint8Array(0)));
});
it('Zip: extra field is written to both headers', () => {
	const zip = new Zlib.Zip();
	const extra = new Uint8Array([0x55, 0x54, 0x03, 0x00, 0x01, 0x02, 0x03]);
	zip.addFile(H.makeText(5000, 2), { filename: new Uint8Array(Buffer.from('with-extra')), extraField: extra });
	zip.addFile(H.makeText(3000, 3), { filename: new Uint8Array(Buffer.from('plain')) });
	const z = zip.compress();
	assert.strictEqual(new DataView(z.buffer).getUint16(28, true), extra.length); // local header's extra length
	assert.ok(H.equal(z.subarray(30 + 10, 30 + 10 + extra.length), extra));
	const files = unzip(z);
	assert.ok(H.equal(files['with-extra'], H.makeText(5000, 2)));
	assert.ok(H.equal(files.plain, H.makeText(3000, 3)));
});

it('Zip: streams written in place give the archive the copying way gives', () => {
	const date = new Date(2024, 0, 2, 3, 4, 6);
	const build = (copy) => {
		const zip = new Zlib.Zip({ comment: new Uint8Array(Buffer.from('archive comment')) });
		const mt = new H.MT(5);
		for (let f = 0; f < 200; ++f) {
			const n = mt.next() % 40000;
			zip.addFile(f % 5 === 4 ? H.makeRandomData(n, f) : H.makeText(n, f), {
				filename: new Uint8Array(Buffer.from('dir/' + f + '.txt')), date,
				extraField: f % 7 === 0 ? new Uint8Array([1, 2, 3, 4]) : undefined,
				comment: f % 11 === 0 ? new Uint8Array(Buffer.from('c' + f)) : undefined,
			});
		}
		Zlib.Zip.copyStreams = copy;
		try {
			return zip.compress();
		} finally {
			Zlib.Zip.copyStreams = false;
		}
	};
	const inPlace = build(false), copied = build(true);
	assert.ok(H.equal(inPlace, copied));
	assert.strictEqual(Object.keys(unzip(inPlace)).length, 200);
});

// ---- batch API ----
it('RawDeflate.compressBatch', () => {
	const ins = [inputs.random.subarray(0, 1000), inputs.sequential, H.makeText(100000, 3), new Uint8Array(0), new Uint8Array([65])];
	const outs = Zlib.RawDeflate.compressBatch(ins, { compressionType: CT.DYNAMIC });
	ins.forEach((d, i) => assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.from(outs[i])), d), 'unit ' + i));
});

// results of 4 MiB and more come in a buffer the addon maps itself (huge pages) and hands over as an
// external ArrayBuffer: the views, a zip archive framed inside such a buffer, and many buffers made and dropped
it('allocInput: page-locked inputs and the one-step path give the bytes of ordinary ones', () => {
	// 40 MiB: above the size from which a batch is written straight into a mapping sized for the worst case
	const n = 40 << 20, ub = 1 << 18;
	const plain = new Uint8Array(n);
	let x = 7;
	for (let i = 0; i < n; ++i) {
		if ((i & 7) === 0) x = (Math.imul(x, 1664525) + 1013904223) >>> 0;
		plain[i] = 97 + ((x >>> (4 * (i & 7))) & 15);
	}
	const locked = G.allocInput(n);
	assert.ok(locked instanceof Uint8Array && locked.length === n && locked.byteOffset === 0);
	locked.set(plain);
	const cut = (a) => { const v = []; for (let o = 0; o < n; o += ub) v.push(a.subarray(o, o + ub)); return v; };
	const a = Zlib.RawDeflate.compressBatch(cut(plain), { compressionType: CT.DYNAMIC });
	const b = Zlib.RawDeflate.compressBatch(cut(locked), { compressionType: CT.DYNAMIC });
	assert.strictEqual(a.length, b.length);
	assert.strictEqual(a[0].buffer, a[a.length - 1].buffer);
	// the ArrayBuffer ends with the last stream: its length is the compressed size, not the mapping's
	assert.strictEqual(a[0].buffer.byteLength, a[a.length - 1].byteOffset + a[a.length - 1].length);
	for (let i = 0; i < a.length; ++i) assert.ok(H.equal(a[i], b[i]), 'unit ' + i);
	for (const i of [0, 1, 77, a.length - 1]) assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.from(a[i])), plain.subarray(i * ub, (i + 1) * ub)));
	// below that size the two-step path runs: same bytes
	const small = Zlib.RawDeflate.compressBatch(cut(plain).slice(0, 8), { compressionType: CT.DYNAMIC });
	for (let i = 0; i < 8; ++i) assert.ok(H.equal(small[i], a[i]), 'two-step unit ' + i);
	// the caller's own output buffer (page-locked or not, with an offset): views of it, same bytes; too small throws
	for (const mk of [(k) => G.allocInput(k), (k) => new Uint8Array(k + 24).subarray(24)]) {
		const out = mk(n);
		const c = Zlib.RawDeflate.compressBatch(cut(locked), { compressionType: CT.DYNAMIC, output: out });
		assert.strictEqual(c.length, a.length);
		assert.strictEqual(c[0].buffer, out.buffer);
		assert.strictEqual(c[0].byteOffset, out.byteOffset);
		for (let i = 0; i < a.length; ++i) assert.ok(H.equal(a[i], c[i]), 'own buffer, unit ' + i);
		const again = Zlib.RawDeflate.compressBatch(cut(plain).slice(0, 8), { compressionType: CT.DYNAMIC, output: out });
		for (let i = 0; i < 8; ++i) assert.ok(H.equal(again[i], a[i]), 'own buffer again, unit ' + i);
	}
	assert.throws(() => Zlib.RawDeflate.compressBatch(cut(plain), { compressionType: CT.DYNAMIC, output: new Uint8Array(1000) }), /output/);
	assert.throws(() => Zlib.RawDeflate.compressBatch(cut(plain), { compressionType: CT.DYNAMIC, output: [1, 2] }), TypeError);
	assert.throws(() => Zlib.RawDeflate.compressBatch(cut(locked), { compressionType: CT.DYNAMIC, output: locked.subarray(1000) }), /overlaps/);
	G.addon.hostTrim();
});

it('large results: externally mapped output buffers', () => {
	const rnd = new Uint8Array(6 << 20);
	let x = 99;
	for (let i = 0; i < rnd.length; i += 4) {
		x = (Math.imul(x, 1664525) + 1013904223) >>> 0;
		rnd[i] = x; rnd[i + 1] = x >>> 8; rnd[i + 2] = x >>> 16; rnd[i + 3] = x >>> 24;
	}
	let lastByte = -1;
	for (let round = 0; round < 6; ++round) {
		const one = new Zlib.RawDeflate(rnd, { compressionType: CT.DYNAMIC }).compress();
		assert.ok(one.length >= rnd.length && one.byteOffset === 0);
		if (round === 0) assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.from(one)), rnd));
		else assert.strictEqual(one[one.length - 1], lastByte);
		lastByte = one[one.length - 1];
	}
	const parts = [];
	for (let i = 0; i < 24; ++i) parts.push(rnd.subarray(i << 18, (i + 1) << 18));
	const outs = Zlib.RawDeflate.compressBatch(parts, { compressionType: CT.FIXED });
	assert.strictEqual(outs[0].buffer, outs[23].buffer);
	assert.ok(outs[0].buffer.byteLength >= 4 << 20);
	parts.forEach((d, i) => assert.ok(H.equal(nodeZlib.inflateRawSync(Buffer.from(outs[i])), d), 'part ' + i));
	const zip = new Zlib.Zip();
	parts.forEach((d, i) => zip.addFile(d, { filename: new Uint8Array(Buffer.from('p' + i)) }));
	const files = unzip(zip.compress());
	parts.forEach((d, i) => assert.ok(H.equal(files['p' + i], d), 'entry ' + i));
	if (global.gc) global.gc();
});

// ---- Node wrapper, compress side (test/nodejs/node-test.js:101-131, node/exports.js:52-69,150-165) ----
it('deflateSync / gzipSync return Buffers that Node inflates', () => {
	const W = require('../zlib_gpu.js');
	const data = inputs['random sequential'];
	const d = W.deflateSync(data);
	assert.ok(Buffer.isBuffer(d));
	assert.ok(H.equal(nodeZlib.inflateSync(d), data));
	const f = W.deflateSync(data, { deflateOption: { compressionType: CT.FIXED }, noBuffer: true });
	assert.ok(f instanceof Uint8Array && !Buffer.isBuffer(f));
	assert.ok(H.equal(nodeZlib.inflateSync(Buffer.from(f)), data));
	assert.ok(H.equal(nodeZlib.gunzipSync(W.gzipSync(data)), data));
	let called = false;
	W.deflate(data, (err, out) => {
		called = true;
		assert.ok(!err);
		assert.ok(H.equal(nodeZlib.inflateSync(out), data));
	});
	assert.strictEqual(called, false); // runs on the next tick
});

process.on('exit', () => console.log(`${passed} tests passed`));
