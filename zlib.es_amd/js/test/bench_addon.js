// Throughput of the N-API path (host buffers, PCIe and JS object creation included).
//   node zlib.es_amd/js/test/bench_addon.js
'use strict';
const { Zlib, deflateChunked, allocInput } = require('../zlib_gpu.js');
const n = 16384, ub = 65536;
const big = new Uint8Array(n * ub);
let x = 12345;
for (let i = 0; i < big.length; ++i) {
	if ((i & 7) === 0) x = (Math.imul(x, 1664525) + 1013904223) >>> 0;
	big[i] = 97 + ((x >>> (4 * (i & 7))) & 15);
}
const views = [], copies = [];
for (let i = 0; i < n; ++i) views.push(big.subarray(i * ub, (i + 1) * ub));
for (let i = 0; i < n; i += 16) copies.push(big.slice(i * ub, (i + 1) * ub));
const asJson = process.argv.includes('--json');
const results = {};
function time(name, bytes, fn) {
	fn();
	let best = Infinity, r;
	for (let k = 0; k < 3; ++k) {
		const t = process.hrtime.bigint();
		r = fn();
		best = Math.min(best, Number(process.hrtime.bigint() - t) / 1e9);
	}
	results[name] = { value: +(bytes / best / 1e9).toFixed(2), unit: 'GB/s', ms: +(best * 1e3).toFixed(1) };
	if (!asJson) console.log(name + ': ' + (bytes / best / 1e9).toFixed(2) + ' GB/s (' + (best * 1e3).toFixed(1) + ' ms)');
	return r;
}
time('compressBatch, 16384 x 64 KiB views of one buffer', n * ub, () => Zlib.RawDeflate.compressBatch(views, { compressionType: 2 }));
{
	// the same input in page-locked memory (allocInput): uploads by DMA instead of through the runtime's staging
	const locked = allocInput(n * ub);
	locked.set(big);
	const lviews = [];
	for (let i = 0; i < n; ++i) lviews.push(locked.subarray(i * ub, (i + 1) * ub));
	time('compressBatch, 16384 x 64 KiB views of one allocInput buffer', n * ub, () => Zlib.RawDeflate.compressBatch(lviews, { compressionType: 2 }));
	// ... and the output into a page-locked buffer the caller keeps from call to call
	const out = allocInput(n * ub * 3 / 4);
	time('compressBatch, allocInput buffer in, kept allocInput buffer out', n * ub, () => Zlib.RawDeflate.compressBatch(lviews, { compressionType: 2, output: out }));
}
time('compressBatch, 1024 separate 64 KiB buffers', copies.length * ub, () => Zlib.RawDeflate.compressBatch(copies, { compressionType: 2 }));
time('deflateChunked, 1 GiB in 1 MiB chunks', n * ub, () => deflateChunked(big, { chunkSize: 1 << 20 }));
time('RawDeflate#compress, one 64 MiB input', 64 << 20, () => new Zlib.RawDeflate(big.subarray(0, 64 << 20)).compress());
if (asJson) console.log(JSON.stringify(Object.assign({ what: 'zlib_gpu.js under Node ' + process.version + ' (N-API addon, host buffers, PCIe and JS object creation included), best of 3' }, results)));
