// Compresses the inputs a job file names through the shim's classes and prints the streams
// (base64, one JSON array on the last line).  Used to make and to re-check the stream fixture.
//   node test/make_streams.js jobs.json      jobs: [{file, kind: 'raw'|'zlib'|'gzip'}]
'use strict';
const fs = require('fs');
const { Zlib } = require('../zlib_gpu.js');
const jobs = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = [];
for (const j of jobs) {
	const data = new Uint8Array(fs.readFileSync(j.file));
	let comp;
	if (j.kind === 'raw') comp = new Zlib.RawDeflate(data, { compressionType: 2 }).compress();
	else if (j.kind === 'zlib') comp = new Zlib.Deflate(data).compress();
	else if (j.kind === 'gzip') comp = new Zlib.Gzip(data, { filename: 'f.bin', flags: { fname: true, fhcrc: true } }).compress();
	else throw new Error('kind ' + j.kind);
	out.push(Buffer.from(comp.buffer, comp.byteOffset, comp.length).toString('base64'));
}
console.log(JSON.stringify(out));
