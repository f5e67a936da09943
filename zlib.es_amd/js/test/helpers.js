'use strict';
// Test data generators with the same statistics as the reference's
// test/browser.es/staff/util.js:87-134 (MT19937 low bytes, i & 0xff, runs of
// incrementing bytes), restated here because the reference tree is not on the
// GPU box.
class MT {
	constructor(seed) {
		this.mt = new Uint32Array(624);
		this.mt[0] = seed >>> 0;
		for (let i = 1; i < 624; i++) {
			const x = this.mt[i - 1] ^ (this.mt[i - 1] >>> 30);
			this.mt[i] = (Math.imul(1812433253, x) + i) >>> 0;
		}
		this.idx = 624;
	}
	next() {
		const mt = this.mt;
		if (this.idx >= 624) {
			for (let k = 0; k < 624; k++) {
				const y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7fffffff);
				mt[k] = mt[(k + 397) % 624] ^ (y >>> 1) ^ (y & 1 ? 0x9908b0df : 0);
			}
			this.idx = 0;
		}
		let v = mt[this.idx++];
		v ^= v >>> 11;
		v ^= (v << 7) & 0x9d2c5680;
		v ^= (v << 15) & 0xefc60000;
		v ^= v >>> 18;
		return v >>> 0;
	}
	nextInt256() {
		return this.next() & 0xff;
	}
}
function makeRandomData(size, seed) {
	const mt = new MT(seed), d = new Uint8Array(size);
	for (let i = 0; i < size; ++i) d[i] = mt.nextInt256();
	return d;
}
function makeSequentialData(size) {
	const d = new Uint8Array(size);
	for (let i = 0; i < size; ++i) d[i] = i & 0xff;
	return d;
}
function makeRandomSequentialData(size, seed) {
	const mt = new MT(seed), d = new Uint8Array(size);
	for (let i = 0; i < size; ) {
		let r1 = mt.nextInt256(), r2 = mt.nextInt256();
		while (r2--) {
			if (i === size) break;
			d[i++] = r1++ & 0xff;
		}
	}
	return d;
}
function makeText(size, seed) {
	const mt = new MT(seed), d = new Uint8Array(size), vocab = [];
	for (let w = 0; w < 2048; ++w) {
		const len = 2 + (mt.next() % 9), word = [];
		for (let k = 0; k < len; ++k) word.push(97 + (mt.next() % 26));
		vocab.push(word);
	}
	for (let i = 0; i < size; ) {
		const u = mt.next() / 4294967296, word = vocab[Math.floor(u * u * u * 2048)];
		for (let k = 0; k < word.length && i < size; ++k) d[i++] = word[k];
		if (i < size) d[i++] = 32;
	}
	return d;
}
function equal(a, b) {
	if (a.length !== b.length) return false;
	for (let i = 0; i < a.length; ++i) if (a[i] !== b[i]) return false;
	return true;
}
// plain table CRC-32 (checker only)
const T = new Uint32Array(256);
for (let i = 0; i < 256; ++i) {
	let c = i;
	for (let k = 0; k < 8; ++k) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1;
	T[i] = c >>> 0;
}
function crc32(d) {
	let c = 0xffffffff;
	for (let i = 0; i < d.length; ++i) c = (c >>> 8) ^ T[(c ^ d[i]) & 0xff];
	return (c ^ 0xffffffff) >>> 0;
}
function adler32(d) {
	let a = 1, b = 0;
	for (let i = 0; i < d.length; ++i) {
		a = (a + d[i]) % 65521;
		b = (b + a) % 65521;
	}
	return ((b << 16) | a) >>> 0;
}
module.exports = { MT, makeRandomData, makeSequentialData, makeRandomSequentialData, makeText, equal, crc32, adler32 };
