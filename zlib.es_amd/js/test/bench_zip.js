// Zip#compress() through the shim on a batch of small files (host buffers, PCIe, JS framing included).
//   node zlib.es_amd/js/test/bench_zip.js [files] [bytes per file]
'use strict';
const { Zlib } = require('../zlib_gpu.js');
const n = +(process.argv[2] || 65536), ub = +(process.argv[3] || 8192);
const big = new Uint8Array(n * ub);
let x = 12345;
for (let i = 0; i < big.length; ++i) {
	if ((i & 7) === 0) x = (Math.imul(x, 1664525) + 1013904223) >>> 0;
	big[i] = 97 + ((x >>> (4 * (i & 7))) & 15);
}
function run() {
	const zip = new Zlib.Zip();
	for (let i = 0; i < n; ++i) {
		const name = 'dir' + (i >> 10) + '/file' + i + '.txt';
		zip.addFile(big.subarray(i * ub, (i + 1) * ub), { filename: Array.from(Buffer.from(name)) });
	}
	const t = process.hrtime.bigint();
	const out = zip.compress();
	return [Number(process.hrtime.bigint() - t) / 1e9, out.length];
}
run();
let best = Infinity, len = 0;
for (let k = 0; k < 3; ++k) {
	const [dt, l] = run();
	best = Math.min(best, dt);
	len = l;
}
console.log(JSON.stringify({ what: 'Zip#compress(), ' + n + ' files of ' + ub + ' B', value: +(n * ub / best / 1e9).toFixed(2), unit: 'GB/s', ms: +(best * 1e3).toFixed(1), archive_bytes: len }));
