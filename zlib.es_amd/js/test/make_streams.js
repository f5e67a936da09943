// Compresses the inputs a job file names through the shim's classes and prints the streams
// (base64, one JSON array on the last line).  Used to make and to re-check the stream fixture.
//   node test/make_streams.js jobs.json      jobs: [{file, kind: 'raw'|'zlib'|'gzip'} | {kind: 'zip', files, names, password}]
'use strict';
const fs = require('fs');
const { Zlib } = require('../zlib_gpu.js');
const jobs = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = [];
for (const j of jobs) {
	let comp;
	if (j.kind === 'zip') {
		// an archive under a password; Math.random (the twelve header bytes per entry, as in src/zip.js:144)
		// is pinned while it is made, so that the same call gives the same archive
		const realRandom = Math.random;
		let st = 20121006;
		Math.random = () => {
			st = (st * 48271) % 2147483647;
			return st / 2147483647;
		};
		try {
			const zip = new Zlib.Zip();
			zip.setPassword(j.password);
			j.files.forEach((f, i) => zip.addFile(new Uint8Array(fs.readFileSync(f)), { filename: new Uint8Array(Buffer.from(j.names[i])), date: new Date(2024, 0, 2, 3, 4, 6) }));
			comp = zip.compress();
		} finally {
			Math.random = realRandom;
		}
		out.push(Buffer.from(comp.buffer, comp.byteOffset, comp.length).toString('base64'));
		continue;
	}
	const data = new Uint8Array(fs.readFileSync(j.file));
	if (j.kind === 'raw') comp = new Zlib.RawDeflate(data, { compressionType: 2 }).compress();
	else if (j.kind === 'zlib') comp = new Zlib.Deflate(data).compress();
	else if (j.kind === 'gzip') comp = new Zlib.Gzip(data, { filename: 'f.bin', flags: { fname: true, fhcrc: true } }).compress();
	else throw new Error('kind ' + j.kind);
	out.push(Buffer.from(comp.buffer, comp.byteOffset, comp.length).toString('base64'));
}
console.log(JSON.stringify(out));
