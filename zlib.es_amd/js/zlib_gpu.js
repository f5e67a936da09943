'use strict';
// JavaScript shim with the class surface of ryunosinfx/zlib.es's compression
// side -- Zlib.RawDeflate / Deflate / Gzip / Zip / CRC32 / Adler32 -- whose
// LZ77 + Huffman + bit packing + CRC-32 + Adler-32 work runs on the GPU through
// the N-API addon (addon.cc -> libzgpu.so).  Container framing stays here, in
// JavaScript, as in the reference (src/deflate.js, src/gzip.js, src/zip.js).
// Inflate stays with the reference's own decoders (not part of this module).
//
// Differences from the reference, each yielding a valid stream where the
// reference emits an invalid or odd one: empty input with NONE gives one empty
// stored block (reference: the untouched 32 KiB buffer, src/rawdeflate.js:106),
// empty input with FIXED carries an end-of-block code (:287,:356), and
// lazy > 0 never emits the overlapping tail tokens of :329.
const path = require('path');
const addon = require('./zgpu_addon.node');
// ZGPU_DEVICES ("all", or "0,1,...") names the GPUs a large batch (Zip.compress(), the chunked helpers,
// compressBatch) is split over inside the library; unset, the calls run on the current device.
if (process.env.ZGPU_DEVICES) addon.init(process.env.ZGPU_LIB || path.join(__dirname, '..', 'libzgpu.so'), process.env.ZGPU_DEVICES);
else addon.init(process.env.ZGPU_LIB || path.join(__dirname, '..', 'libzgpu.so'));

const F_FINAL_EACH = 1, F_SYNC_ALIGN = 2, F_FINAL_LAST = 4, F_PRIME_WINDOW = 8;

function asU8(input) {
	if (input instanceof Uint8Array) return input;
	if (Array.isArray(input)) return new Uint8Array(input);
	if (typeof input === 'string') {
		// Util.stringToByteArray, src/util.js
		const a = new Uint8Array(input.length);
		for (let i = 0; i < input.length; ++i) a[i] = input.charCodeAt(i) & 0xff;
		return a;
	}
	if (ArrayBuffer.isView(input)) return new Uint8Array(input.buffer, input.byteOffset, input.byteLength);
	throw new TypeError('unsupported input');
}

// The first `used` bytes of `body` followed by `extra` more: in place when body's buffer
// has the room (the reference re-wraps result.buffer the same way, src/gzip.js:112-116,
// src/deflate.js:87-93), else in a new buffer.  Always a view at byteOffset 0.
function withTrailerRoom(body, used, extra) {
	if (body.byteOffset === 0 && body.buffer.byteLength >= used + extra) return new Uint8Array(body.buffer, 0, used + extra);
	const out = new Uint8Array(used + extra);
	out.set(body.subarray(0, used));
	return out;
}

class CRC32 {
	// src/crc32.js:14-50
	static calc(data, pos, length) {
		return CRC32.update(data, 0, pos, length);
	}
	static update(data, crc, pos, length) {
		const a = asU8(data);
		const p = typeof pos === 'number' ? pos : 0;
		const l = typeof length === 'number' ? length : a.length;
		const sub = a.subarray(p, p + l); // clamped to the array's end, as the combine below must be
		const c = addon.checksum(sub)[0];
		return addon.crc32Combine(crc >>> 0, c, sub.length) >>> 0;
	}
	static single(num, crc) {
		// one table step, used by ZipCrypto key updates only (src/crc32.js:48-50)
		if (!CRC32.Table) {
			const t = new Uint32Array(256);
			for (let i = 0; i < 256; ++i) {
				let c = i;
				for (let k = 0; k < 8; ++k) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1;
				t[i] = c >>> 0;
			}
			CRC32.Table = t;
		}
		return (CRC32.Table[(num ^ crc) & 0xff] ^ (num >>> 8)) >>> 0;
	}
}

class Adler32 {
	// src/adler32.js:11-37
	static mkHash(array) {
		return Adler32.update(1, array);
	}
	static update(adler, array) {
		const a = asU8(array);
		const h = addon.checksum(a)[1];
		return addon.adler32Combine(adler >>> 0, h, a.length) >>> 0;
	}
}

class RawDeflate {
	// src/rawdeflate.js:81-97
	constructor(input, opt_params) {
		this.compressionType = RawDeflate.CompressionType.DYNAMIC;
		this.lazy = 0;
		this.input = input instanceof Array ? new Uint8Array(input) : input;
		this.op = 0;
		if (opt_params) {
			if (opt_params.lazy) this.lazy = opt_params.lazy;
			if (typeof opt_params.compressionType === 'number') this.compressionType = opt_params.compressionType;
			if (opt_params.outputBuffer)
				this.output = opt_params.outputBuffer instanceof Array ? new Uint8Array(opt_params.outputBuffer) : opt_params.outputBuffer;
			if (typeof opt_params.outputIndex === 'number') this.op = opt_params.outputIndex;
		}
		if (!this.output) this.output = new Uint8Array(0x8000);
	}
	// src/rawdeflate.js:102-124
	compress() {
		const input = this.input;
		switch (this.compressionType) {
			case RawDeflate.CompressionType.NONE:
				this.output = this.makeNocompressBlock(input, true);
				break;
			case RawDeflate.CompressionType.FIXED:
				this.output = this.makeFixedHuffmanBlock(input, true);
				break;
			case RawDeflate.CompressionType.DYNAMIC:
				this.output = this.makeDynamicHuffmanBlock(input, true);
				break;
			default:
				throw 'invalid compression type';
		}
		this.op = this.output.length;
		return this.output;
	}
	// The three block makers stay real call sites (the reference's tests spy on
	// them, test/browser.es/code-path-test.js:22-24); each hands its block to
	// the GPU and appends the result after the bytes already in the output.
	_gpuBlock(blockArray, ctype, isFinalBlock) {
		// a container that needs the input's checksum (Deflate: Adler-32, Gzip: CRC-32) asks for it
		// with the same engine call, so the input crosses PCIe once
		const want = this._want || 0;
		const r = addon.deflateBatch([asU8(blockArray)], ctype, this.lazy | 0, isFinalBlock ? F_FINAL_EACH : F_SYNC_ALIGN, want === 1, want === 2);
		if (want === 1) this._crc32 = r.crc32[0];
		if (want === 2) this._adler32 = r.adler32[0];
		const body = r.outputs[0];
		if (this.op === 0 && body.byteOffset === 0 && body.byteLength === body.buffer.byteLength) {
			// nothing before it: the addon's buffer already is a view at byteOffset 0 of its own buffer
			this.op = body.length;
			this.output = body;
			return body;
		}
		// view at byteOffset 0 of its own buffer, with room for the 8 trailer bytes the container
		// classes append in place when the buffer allows (src/gzip.js:112-116, src/deflate.js:87-93)
		const out = new Uint8Array(new ArrayBuffer(this.op + body.length + 8), 0, this.op + body.length);
		out.set(this.output.subarray(0, this.op));
		out.set(body, this.op);
		this.op = out.length;
		this.output = out;
		return out;
	}
	makeNocompressBlock(blockArray, isFinalBlock) {
		return this._gpuBlock(blockArray, RawDeflate.CompressionType.NONE, isFinalBlock);
	}
	makeFixedHuffmanBlock(blockArray, isFinalBlock) {
		return this._gpuBlock(blockArray, RawDeflate.CompressionType.FIXED, isFinalBlock);
	}
	makeDynamicHuffmanBlock(blockArray, isFinalBlock) {
		return this._gpuBlock(blockArray, RawDeflate.CompressionType.DYNAMIC, isFinalBlock);
	}
	// One engine call for many independent inputs: what the per-file loop of
	// Zip.compress() (src/zip.js:119-137) becomes.
	// opt_params.output (not in the reference): a Uint8Array the streams are written into back to back --
	// the returned arrays are then views of it, valid until the caller reuses it.  A caller that compresses
	// batch after batch keeps one such buffer (from allocInput: page-locked) instead of getting a new
	// ArrayBuffer of fresh pages per call; too small a buffer throws.
	static compressBatch(inputs, opt_params) {
		const o = opt_params || {};
		const ct = typeof o.compressionType === 'number' ? o.compressionType : RawDeflate.CompressionType.DYNAMIC;
		if (ct !== 0 && ct !== 1 && ct !== 2) throw 'invalid compression type';
		if (o.output !== undefined) {
			if (!(o.output instanceof Uint8Array)) throw new TypeError('output: a Uint8Array');
			return addon.deflateBatch(inputs.map(asU8), ct, o.lazy | 0, F_FINAL_EACH, false, false, o.output).outputs;
		}
		return addon.deflateBatch(inputs.map(asU8), ct, o.lazy | 0, F_FINAL_EACH, false, false).outputs;
	}
}
// UNCOMPRESSED: the bundle's alias of NONE (reference src/zlib.es.js:111-117)
RawDeflate.CompressionType = { UNCOMPRESSED: 0, NONE: 0, FIXED: 1, DYNAMIC: 2, RESERVED: 3 };

class Deflate {
	// src/deflate.js:20-30
	constructor(input, opt_params) {
		this.input = input;
		this.output = new Uint8Array(Deflate.DefaultBufferSize);
		this.compressionType = RawDeflate.CompressionType.DYNAMIC;
		const rawDeflateOption = {};
		if (opt_params || !(opt_params = {})) if (typeof opt_params.compressionType === 'number') this.compressionType = opt_params.compressionType;
		for (const prop in opt_params) rawDeflateOption[prop] = opt_params[prop];
		rawDeflateOption.outputBuffer = this.output;
		this.rawDeflate = new RawDeflate(this.input, rawDeflateOption);
	}
	static compress(input, opt_params) {
		return new Deflate(input, opt_params).compress();
	}
	// src/deflate.js:44-99
	compress() {
		const output = this.output;
		let pos = 0;
		const cmf = (7 << 4) | 8; // CINFO for the 32 KiB window, CM = deflate
		output[pos++] = cmf;
		let flevel;
		switch (this.compressionType) {
			case RawDeflate.CompressionType.NONE: flevel = 0; break;
			case RawDeflate.CompressionType.FIXED: flevel = 1; break;
			case RawDeflate.CompressionType.DYNAMIC: flevel = 2; break;
			default: throw new Error('unsupported compression type');
		}
		let flg = flevel << 6;
		flg |= 31 - ((cmf * 256 + flg) % 31);
		output[pos++] = flg;
		this.rawDeflate.op = pos;
		this.rawDeflate._want = 2; // Adler-32 of the input from the same call (src/deflate.js:83 is a separate pass)
		const body = this.rawDeflate.compress();
		const adler = this.rawDeflate._adler32 !== undefined ? this.rawDeflate._adler32 : Adler32.mkHash(this.input);
		let pos2 = body.length;
		const out = withTrailerRoom(body, pos2, 4);
		out[pos2++] = (adler >> 24) & 0xff;
		out[pos2++] = (adler >> 16) & 0xff;
		out[pos2++] = (adler >> 8) & 0xff;
		out[pos2++] = adler & 0xff;
		this.output = out;
		return out;
	}
}
Deflate.DefaultBufferSize = 0x8000;
Deflate.CompressionType = RawDeflate.CompressionType;

class Gzip {
	// src/gzip.js:43-55
	constructor(input, opt_params) {
		this.input = input;
		this.ip = 0;
		this.op = 0;
		this.flags = {};
		if (opt_params) {
			if (opt_params.flags) this.flags = opt_params.flags;
			if (typeof opt_params.filename === 'string') this.filename = opt_params.filename;
			if (typeof opt_params.comment === 'string') this.comment = opt_params.comment;
			if (opt_params.deflateOptions) this.deflateOptions = opt_params.deflateOptions;
		}
		if (!this.deflateOptions) this.deflateOptions = {};
	}
	// src/gzip.js:60-131
	compress() {
		const filename = this.filename, comment = this.comment;
		const head = [];
		head.push(0x1f, 0x8b, 8);
		let flg = 0;
		if (this.flags.fname) flg |= Gzip.FlagsMask.FNAME;
		if (this.flags.fcomment) flg |= Gzip.FlagsMask.FCOMMENT;
		if (this.flags.fhcrc) flg |= Gzip.FlagsMask.FHCRC;
		head.push(flg);
		const mtime = ((Date.now ? Date.now() : +new Date()) / 1000) | 0;
		head.push(mtime & 0xff, (mtime >>> 8) & 0xff, (mtime >>> 16) & 0xff, (mtime >>> 24) & 0xff);
		head.push(0, Gzip.OperatingSystem.UNKNOWN);
		const chars = (s) => {
			for (let i = 0, il = s.length; i < il; ++i) {
				const c = s.charCodeAt(i);
				if (c > 0xff) head.push((c >>> 8) & 0xff);
				head.push(c & 0xff);
			}
			head.push(0);
		};
		if (this.flags.fname !== void 0) chars(filename);
		if (this.flags.comment) chars(comment);
		if (this.flags.fhcrc) {
			const crc16 = CRC32.calc(new Uint8Array(head), 0, head.length) & 0xffff;
			head.push(crc16 & 0xff, (crc16 >>> 8) & 0xff);
		}
		const output = new Uint8Array(Math.max(Gzip.DefaultBufferSize, head.length));
		output.set(head);
		this.deflateOptions.outputBuffer = output;
		this.deflateOptions.outputIndex = head.length;
		const rawdeflate = new RawDeflate(this.input, this.deflateOptions);
		rawdeflate._want = 1; // CRC-32 of the input from the same call (src/gzip.js:117 is a separate pass)
		const body = rawdeflate.compress();
		let op2 = rawdeflate.op;
		const out = withTrailerRoom(body, op2, 8);
		const crc32 = rawdeflate._crc32 !== undefined ? rawdeflate._crc32 : CRC32.calc(this.input);
		out[op2++] = crc32 & 0xff;
		out[op2++] = (crc32 >>> 8) & 0xff;
		out[op2++] = (crc32 >>> 16) & 0xff;
		out[op2++] = (crc32 >>> 24) & 0xff;
		const il = this.input.length;
		out[op2++] = il & 0xff;
		out[op2++] = (il >>> 8) & 0xff;
		out[op2++] = (il >>> 16) & 0xff;
		out[op2++] = (il >>> 24) & 0xff;
		this.output = out;
		return out;
	}
}
Gzip.DefaultBufferSize = 0x8000;
Gzip.OperatingSystem = { FAT: 0, AMIGA: 1, VMS: 2, UNIX: 3, VM_CMS: 4, ATARI_TOS: 5, HPFS: 6, MACINTOSH: 7, Z_SYSTEM: 8, CP_M: 9, TOPS_20: 10, NTFS: 11, QDOS: 12, ACORN_RISCOS: 13, UNKNOWN: 255 };
Gzip.FlagsMask = { FTEXT: 0x01, FHCRC: 0x02, FEXTRA: 0x04, FNAME: 0x08, FCOMMENT: 0x10 };

class Zip {
	// src/zip.js:46-58
	constructor(opt_params = {}) {
		this.files = [];
		this.comment = opt_params.comment;
		this.password = void 0;
	}
	// src/zip.js:63-91
	addFile(input, opt_params = {}) {
		let compressed = false;
		let crc32 = 0;
		let buffer = input instanceof Array ? new Uint8Array(input) : input;
		if (typeof opt_params.compressionMethod !== 'number') opt_params.compressionMethod = Zip.CompressionMethod.DEFLATE;
		if (opt_params.compress)
			switch (opt_params.compressionMethod) {
				case Zip.CompressionMethod.STORE:
					break;
				case Zip.CompressionMethod.DEFLATE:
					crc32 = CRC32.calc(buffer);
					buffer = this.deflateWithOption(buffer, opt_params);
					compressed = true;
					break;
				default:
					throw new Error(`unknown compression method:${opt_params.compressionMethod}`);
			}
		this.files.push({ buffer, option: opt_params, compressed, encrypted: false, size: input.length, crc32 });
	}
	setPassword(password) {
		this.password = password;
	}
	deflateWithOption(input, opt_params) {
		return new RawDeflate(input, opt_params.deflateOption).compress();
	}
	// The reference's first loop (src/zip.js:119-137), batched: CRC-32 of every
	// pending file and DEFLATE of the DEFLATE ones in one engine call per
	// (compressionType, lazy) group.
	_compressPending() {
		const groups = new Map();
		for (const file of this.files) {
			if (file.compressed) continue;
			const m = file.option.compressionMethod;
			if (m !== Zip.CompressionMethod.STORE && m !== Zip.CompressionMethod.DEFLATE) throw new Error(`unknown compression method:${m}`);
			const d = file.option.deflateOption || {};
			const ct = m === Zip.CompressionMethod.DEFLATE ? (typeof d.compressionType === 'number' ? d.compressionType : 2) : 0;
			if (ct !== 0 && ct !== 1 && ct !== 2) throw 'invalid compression type';
			const key = `${m}:${ct}:${d.lazy | 0}`;
			if (!groups.has(key)) groups.set(key, { m, ct, lazy: d.lazy | 0, files: [] });
			groups.get(key).files.push(file);
		}
		for (const g of groups.values()) {
			const r = addon.deflateBatch(g.files.map((f) => asU8(f.buffer)), g.ct, g.lazy, F_FINAL_EACH, true, false);
			g.files.forEach((f, i) => {
				f.crc32 = r.crc32[i];
				if (g.m === Zip.CompressionMethod.DEFLATE) {
					f.buffer = r.outputs[i];
					f.compressed = true;
				}
			});
		}
	}
	// The usual archive -- every file still to be compressed, all with the same deflate options, no
	// password: one engine call that leaves room for each file's local header in front of its stream and
	// for the central directory behind the last, so the compressed bytes are written once, in place.
	// Returns the archive buffer with the streams in it, or null when the archive is not of that kind.
	_compressInPlace() {
		const files = this.files;
		if (!files.length || this.password !== void 0 || Zip.copyStreams) return null;
		let ct = -1, lazy = 0, tail = 22 + (this.comment ? this.comment.length : 0);
		const gaps = new Uint32Array(files.length);
		for (let i = 0; i < files.length; ++i) {
			const f = files[i], opt = f.option;
			if (f.compressed || opt.password !== void 0 || opt.compressionMethod !== Zip.CompressionMethod.DEFLATE) return null;
			const d = opt.deflateOption || {};
			const c = typeof d.compressionType === 'number' ? d.compressionType : 2;
			if (c !== 1 && c !== 2) return null;
			if (i === 0) {
				ct = c;
				lazy = d.lazy | 0;
			} else if (c !== ct || (d.lazy | 0) !== lazy) return null;
			const nameLen = opt.filename ? opt.filename.length : 0, extraLen = opt.extraField ? opt.extraField.length : 0;
			gaps[i] = 30 + nameLen + extraLen;
			tail += 46 + nameLen + extraLen + (opt.comment ? opt.comment.length : 0);
		}
		const r = addon.deflateBatch(files.map((f) => asU8(f.buffer)), ct, lazy, F_FINAL_EACH, true, false, gaps, tail);
		files.forEach((f, i) => {
			f.crc32 = r.crc32[i];
			f.buffer = r.outputs[i];
			f.compressed = true;
		});
		return new Uint8Array(r.buffer);
	}
	// src/zip.js:98-274
	compress() {
		const inPlace = this._compressInPlace();
		this._compressPending();
		const files = this.files;
		let localFileSize = 0, centralDirectorySize = 0;
		const fileCount = files.length;
		for (let i = 0; i < fileCount; ++i) {
			const file = files[i], opt = file.option;
			const filenameLength = opt.filename ? opt.filename.length : 0;
			const commentLength = opt.comment ? opt.comment.length : 0;
			if (opt.password !== void 0 || this.password !== void 0) {
				const key = this.createEncryptionKey(opt.password || this.password);
				const len = file.buffer.length + 12;
				const buffer = new Uint8Array(len);
				buffer.set(file.buffer, 12);
				for (let j = 0; j < 12; ++j) buffer[j] = this.encode(key, i === 11 ? file.crc32 & 0xff : (Math.random() * 256) | 0);
				for (let j = 12; j < len; ++j) buffer[j] = this.encode(key, buffer[j]);
				file.buffer = buffer;
			}
			const extraLength = opt.extraField ? opt.extraField.length : 0;
			localFileSize += 30 + filenameLength + extraLength + file.buffer.length;
			centralDirectorySize += 46 + filenameLength + extraLength + commentLength;
		}
		const endOfCentralDirectorySize = 22 + (this.comment ? this.comment.length : 0);
		const archiveSize = localFileSize + centralDirectorySize + endOfCentralDirectorySize;
		if (inPlace && inPlace.length !== archiveSize) throw new Error('archive size');
		const output = inPlace || new Uint8Array(archiveSize);
		let op1 = 0, op2 = localFileSize, op3 = op2 + centralDirectorySize;
		const both = (v) => {
			output[op1++] = output[op2++] = v & 0xff;
		};
		let now; // one clock reading serves the files without a date of their own (the reference reads it per file)
		for (const file of files) {
			const opt = file.option;
			const filenameLength = opt.filename ? opt.filename.length : 0;
			const extraFieldLength = opt.extraField ? opt.extraField.length : 0;
			const commentLength = opt.comment ? opt.comment.length : 0;
			const offset = op1;
			for (let k = 0; k < 4; ++k) output[op1++] = Zip.LocalFileHeaderSignature[k];
			for (let k = 0; k < 4; ++k) output[op2++] = Zip.FileHeaderSignature[k];
			const needVersion = 20;
			output[op2++] = needVersion & 0xff;
			output[op2++] = opt.os || Zip.OperatingSystem.MSDOS;
			both(needVersion);
			both(needVersion >> 8);
			let flags = 0;
			if (opt.password || this.password) flags |= Zip.Flags.ENCRYPT;
			both(flags);
			both(flags >> 8);
			const compressionMethod = opt.compressionMethod;
			both(compressionMethod);
			both(compressionMethod >> 8);
			const date = opt.date || now || (now = new Date());
			both(((date.getMinutes() & 0x7) << 5) | ((date.getSeconds() / 2) | 0));
			both((date.getHours() << 3) | (date.getMinutes() >> 3));
			both((((date.getMonth() + 1) & 0x7) << 5) | date.getDate());
			both((((date.getFullYear() - 1980) & 0x7f) << 1) | ((date.getMonth() + 1) >> 3));
			const crc32 = file.crc32;
			both(crc32); both(crc32 >> 8); both(crc32 >> 16); both(crc32 >> 24);
			const size = file.buffer.length;
			both(size); both(size >> 8); both(size >> 16); both(size >> 24);
			const plainSize = file.size;
			both(plainSize); both(plainSize >> 8); both(plainSize >> 16); both(plainSize >> 24);
			both(filenameLength); both(filenameLength >> 8);
			both(extraFieldLength); both(extraFieldLength >> 8);
			output[op2++] = commentLength & 0xff;
			output[op2++] = (commentLength >> 8) & 0xff;
			for (let k = 0; k < 8; ++k) output[op2++] = 0; // disk number, internal + external attributes
			output[op2++] = offset & 0xff;
			output[op2++] = (offset >> 8) & 0xff;
			output[op2++] = (offset >> 16) & 0xff;
			output[op2++] = (offset >> 24) & 0xff;
			if (opt.filename) {
				output.set(opt.filename, op1);
				output.set(opt.filename, op2);
				op1 += filenameLength;
				op2 += filenameLength;
			}
			// extra field after the name in both headers (src/zip.js:231-236)
			if (opt.extraField) {
				output.set(opt.extraField, op1);
				output.set(opt.extraField, op2);
				op1 += extraFieldLength;
				op2 += extraFieldLength;
			}
			if (opt.comment) {
				output.set(opt.comment, op2);
				op2 += commentLength;
			}
			if (!inPlace) output.set(file.buffer, op1);  // in place: the stream is already there
			else if (file.buffer.byteOffset !== op1) throw new Error('stream position');
			op1 += file.buffer.length;
		}
		for (let k = 0; k < 4; ++k) output[op3++] = Zip.CentralDirectorySignature[k];
		for (let k = 0; k < 4; ++k) output[op3++] = 0;
		output[op3++] = fileCount & 0xff;
		output[op3++] = (fileCount >> 8) & 0xff;
		output[op3++] = fileCount & 0xff;
		output[op3++] = (fileCount >> 8) & 0xff;
		for (const v of [centralDirectorySize, localFileSize]) {
			output[op3++] = v & 0xff;
			output[op3++] = (v >> 8) & 0xff;
			output[op3++] = (v >> 16) & 0xff;
			output[op3++] = (v >> 24) & 0xff;
		}
		const commentLength = this.comment ? this.comment.length : 0;
		output[op3++] = commentLength & 0xff;
		output[op3++] = (commentLength >> 8) & 0xff;
		if (this.comment) output.set(this.comment, op3);
		return output;
	}
	// ZipCrypto, src/zip.js:287-318 (host arithmetic around CRC32.single)
	getByte(key) {
		const tmp = (key[2] & 0xffff) | 2;
		return ((tmp * (tmp ^ 1)) >> 8) & 0xff;
	}
	encode(key, n) {
		const tmp = this.getByte(key);
		this.updateKeys(key, n);
		return tmp ^ n;
	}
	updateKeys(key, n) {
		key[0] = CRC32.single(key[0], n);
		key[1] = ((((((key[1] + (key[0] & 0xff)) * 20173) >>> 0) * 6681) >>> 0) + 1) >>> 0;
		key[2] = CRC32.single(key[2], key[1] >>> 24);
	}
	createEncryptionKey(password) {
		const key = new Uint32Array([305419896, 591751049, 878082192]);
		for (let i = 0, il = password.length; i < il; ++i) this.updateKeys(key, password[i] & 0xff);
		return key;
	}
}
Zip.CompressionMethod = { STORE: 0, DEFLATE: 8 };
Zip.copyStreams = false; // true: Zip#compress() always copies the compressed streams into the archive (the reference's way; for tests)
Zip.OperatingSystem = { MSDOS: 0, UNIX: 3, MACINTOSH: 7 };
Zip.Flags = { ENCRYPT: 0x0001, DESCRIPTOR: 0x0008, UTF8: 0x0800 };
Zip.FileHeaderSignature = [0x50, 0x4b, 0x01, 0x02];
Zip.LocalFileHeaderSignature = [0x50, 0x4b, 0x03, 0x04];
Zip.CentralDirectorySignature = [0x50, 0x4b, 0x05, 0x06];

// chunked single-member gzip / zlib bodies: the input is cut into chunks
// compressed side by side, each ends byte-aligned (empty stored block), the last
// carries BFINAL; whole-stream CRC-32 / Adler-32 come from the per-chunk values by
// combine.  A chunk may match into the 32 KiB before it unless opt.independent is
// set (then every chunk decodes without its predecessors, at a slightly larger size).
function deflateChunked(input, opt) {
	const o = opt || {};
	const a = asU8(input);
	const chunk = o.chunkSize || (1 << 20);
	const ct = typeof o.compressionType === 'number' ? o.compressionType : 2;
	const parts = [];
	for (let p = 0; p < a.length || p === 0; p += chunk) {
		parts.push(a.subarray(p, Math.min(a.length, p + chunk)));
		if (a.length === 0) break;
	}
	const flags = F_SYNC_ALIGN | F_FINAL_LAST | (o.independent ? 0 : F_PRIME_WINDOW);
	const r = addon.deflateBatch(parts, ct, o.lazy | 0, flags, true, true);
	let crc = r.crc32[0], adler = r.adler32[0];
	for (let i = 1; i < parts.length; ++i) {
		crc = addon.crc32Combine(crc, r.crc32[i], parts[i].length);
		adler = addon.adler32Combine(adler, r.adler32[i], parts[i].length);
	}
	return { parts: r.outputs, crc32: crc >>> 0, adler32: adler >>> 0 };
}

function gzipChunked(input, opt) {
	const o = opt || {};
	const a = asU8(input);
	const d = deflateChunked(a, o);
	const mtime = typeof o.mtime === 'number' ? o.mtime : ((Date.now() / 1000) | 0);
	let total = 10 + 8;
	for (const p of d.parts) total += p.length;
	const out = new Uint8Array(total);
	out.set([0x1f, 0x8b, 8, 0, mtime & 0xff, (mtime >>> 8) & 0xff, (mtime >>> 16) & 0xff, (mtime >>> 24) & 0xff, 0, 255]);
	let op = 10;
	for (const p of d.parts) {
		out.set(p, op);
		op += p.length;
	}
	const il = a.length >>> 0;
	out.set([d.crc32 & 0xff, (d.crc32 >>> 8) & 0xff, (d.crc32 >>> 16) & 0xff, (d.crc32 >>> 24) & 0xff, il & 0xff, (il >>> 8) & 0xff, (il >>> 16) & 0xff, (il >>> 24) & 0xff], op);
	return out;
}

// One zlib stream (RFC 1950 framing as src/deflate.js:44-99 writes it: 78 xx header, body,
// big-endian Adler-32) whose body is made of chunks compressed side by side; the Adler-32 of the
// whole input comes from the chunks' values by combine.
function zlibChunked(input, opt) {
	const o = opt || {};
	const a = asU8(input);
	const d = deflateChunked(a, o);
	const ct = typeof o.compressionType === 'number' ? o.compressionType : 2;
	const cmf = (7 << 4) | 8;
	let flg = ct << 6;
	flg |= 31 - ((cmf * 256 + flg) % 31);
	let total = 2 + 4;
	for (const p of d.parts) total += p.length;
	const out = new Uint8Array(total);
	out[0] = cmf;
	out[1] = flg;
	let op = 2;
	for (const p of d.parts) {
		out.set(p, op);
		op += p.length;
	}
	out.set([(d.adler32 >>> 24) & 0xff, (d.adler32 >>> 16) & 0xff, (d.adler32 >>> 8) & 0xff, d.adler32 & 0xff], op);
	return out;
}

// compress side of the Node wrapper (reference node/exports.js:28-69,127-165):
// deflateSync takes {deflateOption, noBuffer}; gzipSync ignores every option
// but noBuffer; the callback forms run on the next tick and pass (error, data).
function deflateSync(buffer, optParams) {
	const o = optParams || {};
	const out = new Deflate(buffer, o.deflateOption || {}).compress();
	return o.noBuffer ? out : Buffer.from(out);
}
function gzipSync(buffer, optParams) {
	const out = new Gzip(buffer).compress();
	return optParams && optParams.noBuffer ? out : Buffer.from(out);
}
function nextTickForm(fn) {
	return function (buffer, callback, optParams) {
		process.nextTick(() => {
			let error, data;
			try {
				data = fn(buffer, optParams);
			} catch (e) {
				error = e;
			}
			callback(error, data);
		});
	};
}
const deflate = nextTickForm(deflateSync);
const gzip = nextTickForm(gzipSync);

// allocInput(n): a Uint8Array of n bytes in page-locked memory (not part of the reference's surface).  Inputs
// that live in it -- as a whole or as subarrays -- travel to the GPU by DMA at the full PCIe rate; any other
// typed array works too, through the runtime's staging.  Freed when the array's buffer is collected.
function allocInput(n) {
	return new Uint8Array(addon.allocInput(n));
}

const Zlib = { RawDeflate, Deflate, Gzip, Zip, CRC32, Adler32 };
module.exports = {
	Zlib, RawDeflate, Deflate, Gzip, Zip, CRC32, Adler32, deflateChunked, gzipChunked, zlibChunked,
	deflate, deflateSync, gzip, gzipSync, allocInput, addon,
};
