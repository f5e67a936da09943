"""Host-side mirror of the reference's compression classes, backed by the GPU.

Same names, option objects and call structure as the reference
(src/rawdeflate.js, src/deflate.js, src/gzip.js, src/zip.js, src/crc32.js,
src/adler32.js) so that tests written against the reference read the same
here.  Every byte of LZ77 / Huffman / bit packing / CRC-32 / Adler-32 work is
done by libzgpu.so on the GPU; this file only frames containers.

Deliberate differences from the reference (each produces a *valid* stream
where the reference emits an invalid or odd one):
  * empty input + NONE yields one empty stored block instead of the untouched
    32 KiB buffer (src/rawdeflate.js:106,123);
  * empty input + FIXED carries an end-of-block code (the reference drops it,
    src/rawdeflate.js:287,356);
  * ``lazy`` > 0 never emits the overlapping tail tokens of
    src/rawdeflate.js:329.
"""
import random as _random
import time

import numpy as np

from . import _native as N


def _as_u8(data):
    if isinstance(data, str):  # Util.stringToByteArray, src/util.js
        return np.frombuffer(bytes(ord(c) & 0xFF for c in data), dtype=np.uint8)
    if isinstance(data, (list, tuple)):
        return np.asarray(data, dtype=np.uint8)
    return N._u8(data)


class _hybrid:
    """Lets ``Deflate.compress(input, opt)`` (static, src/deflate.js:37) and
    ``Deflate(input, opt).compress()`` (:44) share one name."""

    def __init__(self, static_fn, inst_fn):
        self.static_fn, self.inst_fn = static_fn, inst_fn

    def __get__(self, obj, cls):
        if obj is None:
            return lambda *a, **k: self.static_fn(cls, *a, **k)
        return lambda *a, **k: self.inst_fn(obj, *a, **k)


class CRC32:
    """src/crc32.js:6-50"""
    _table = None

    @staticmethod
    def calc(data, pos=None, length=None):
        return CRC32.update(data, 0, pos, length)

    @staticmethod
    def update(data, crc, pos=None, length=None):
        a = _as_u8(data)
        pos = 0 if pos is None else pos
        length = a.size if length is None else length  # the reference's `length` counts from pos
        return N.crc32_update(a[pos:pos + length], crc & 0xFFFFFFFF)  # the slice is clamped, and so is the combine

    @staticmethod
    def single(num, crc):
        # one table step (src/crc32.js:48-50); used by ZipCrypto key updates only
        if CRC32._table is None:
            t = []
            for i in range(256):
                c = i
                for _ in range(8):
                    c = (0xEDB88320 ^ (c >> 1)) if (c & 1) else (c >> 1)
                t.append(c)
            CRC32._table = t
        return (CRC32._table[(num ^ crc) & 0xFF] ^ (num >> 8)) & 0xFFFFFFFF


class Adler32:
    """src/adler32.js:5-45"""

    @staticmethod
    def mkHash(array):
        return Adler32.update(1, array)

    @staticmethod
    def update(adler, array):
        return N.adler32_update(_as_u8(array), adler & 0xFFFFFFFF)


class RawDeflate:
    """src/rawdeflate.js:6-606"""

    class CompressionType:
        NONE, FIXED, DYNAMIC, RESERVED = 0, 1, 2, 3
        UNCOMPRESSED = 0  # the bundle's alias of NONE (src/zlib.es.js:111-117)

    def __init__(self, input, opt_params=None):
        self.compressionType = RawDeflate.CompressionType.DYNAMIC
        self.lazy = 0
        self.input = _as_u8(input)
        self.op = 0
        self.output = None
        if opt_params:
            if opt_params.get("lazy"):
                self.lazy = opt_params["lazy"]
            if isinstance(opt_params.get("compressionType"), int):
                self.compressionType = opt_params["compressionType"]
            if opt_params.get("outputBuffer") is not None:
                self.output = _as_u8(opt_params["outputBuffer"])
            if isinstance(opt_params.get("outputIndex"), int):
                self.op = opt_params["outputIndex"]
        if self.output is None:
            self.output = np.zeros(0x8000, dtype=np.uint8)

    # --- block makers: kept as the executed call sites (the reference's tests
    # spy on them, test/browser.es/code-path-test.js:22-24) -------------------
    def _run(self, block, ctype, is_final):
        flags = N.F_FINAL_EACH if is_final else N.F_SYNC_ALIGN
        # a container asks for the input's checksum with the same engine call (one upload)
        want = getattr(self, "_want", 0)
        out, crc, adler = N.deflate_batch([block], ctype, self.lazy, flags, want_crc=want == 1, want_adler=want == 2)
        if want == 1:
            self._crc32 = crc[0]
        if want == 2:
            self._adler32 = adler[0]
        head = self.output[: self.op]
        self.output = np.concatenate([head, np.frombuffer(out[0], dtype=np.uint8)])
        self.op = self.output.size
        return self.output

    def makeNocompressBlock(self, blockArray, isFinalBlock):
        return self._run(blockArray, RawDeflate.CompressionType.NONE, isFinalBlock)

    def makeFixedHuffmanBlock(self, blockArray, isFinalBlock):
        return self._run(blockArray, RawDeflate.CompressionType.FIXED, isFinalBlock)

    def makeDynamicHuffmanBlock(self, blockArray, isFinalBlock):
        return self._run(blockArray, RawDeflate.CompressionType.DYNAMIC, isFinalBlock)

    def compress(self):
        ct = self.compressionType
        if ct == RawDeflate.CompressionType.NONE:
            self.makeNocompressBlock(self.input, True)
        elif ct == RawDeflate.CompressionType.FIXED:
            self.makeFixedHuffmanBlock(self.input, True)
        elif ct == RawDeflate.CompressionType.DYNAMIC:
            self.makeDynamicHuffmanBlock(self.input, True)
        else:
            raise ValueError("invalid compression type")  # thrown as a string at :121
        return self.output

    @staticmethod
    def compressBatch(inputs, opt_params=None):
        """One engine call for many independent inputs (what Zip.compress()'s
        per-file loop, src/zip.js:119-137, becomes)."""
        opt = opt_params or {}
        ct = opt.get("compressionType", RawDeflate.CompressionType.DYNAMIC)
        if ct not in (0, 1, 2):
            raise ValueError("invalid compression type")
        out, _, _ = N.deflate_batch([_as_u8(i) for i in inputs], ct, opt.get("lazy", 0) or 0,
                                    N.F_FINAL_EACH)
        return out


class Deflate:
    """zlib container, src/deflate.js:8-100"""
    CompressionType = RawDeflate.CompressionType
    DefaultBufferSize = 0x8000

    def __init__(self, input, opt_params=None):
        self.input = _as_u8(input)
        self.output = np.zeros(Deflate.DefaultBufferSize, dtype=np.uint8)
        self.compressionType = RawDeflate.CompressionType.DYNAMIC
        opt_params = opt_params or {}
        if isinstance(opt_params.get("compressionType"), int):
            self.compressionType = opt_params["compressionType"]
        raw_opt = dict(opt_params)
        raw_opt["outputBuffer"] = self.output
        self.rawDeflate = RawDeflate(self.input, raw_opt)

    def _compress(self):
        cmf = (7 << 4) | 8  # CINFO for a 32 KiB window, CM = deflate (:49-57)
        if self.compressionType not in (0, 1, 2):
            raise ValueError("unsupported compression type")  # :73
        flg = self.compressionType << 6  # FLEVEL = compressionType (:62-71)
        flg |= 31 - ((cmf * 256 + flg) % 31)
        self.output[0], self.output[1] = cmf, flg
        self.rawDeflate.output = self.output
        self.rawDeflate.op = 2
        self.rawDeflate._want = 2  # Adler-32 of the input (:83) from the same call
        body = self.rawDeflate.compress()  # :84-85
        adler = getattr(self.rawDeflate, "_adler32", None)
        if adler is None:  # a RawDeflate whose block makers were replaced: a checksum pass of its own (:83)
            adler = Adler32.mkHash(self.input)
        trailer = np.array([(adler >> 24) & 0xFF, (adler >> 16) & 0xFF, (adler >> 8) & 0xFF, adler & 0xFF],
                           dtype=np.uint8)  # :94-97
        self.output = np.concatenate([body, trailer])
        return self.output

    compress = _hybrid(lambda cls, input, opt_params=None: cls(input, opt_params).compress(),
                       lambda self: self._compress())


class Gzip:
    """src/gzip.js:6-132"""
    DefaultBufferSize = 0x8000

    class OperatingSystem:
        FAT, AMIGA, VMS, UNIX, VM_CMS, ATARI_TOS, HPFS, MACINTOSH = range(8)
        Z_SYSTEM, CP_M, TOPS_20, NTFS, QDOS, ACORN_RISCOS, UNKNOWN = 8, 9, 10, 11, 12, 13, 255

    class FlagsMask:
        FTEXT, FHCRC, FEXTRA, FNAME, FCOMMENT = 0x01, 0x02, 0x04, 0x08, 0x10

    def __init__(self, input, opt_params=None):
        self.input = _as_u8(input)
        self.ip = 0
        self.op = 0
        self.flags = {}
        self.filename = None
        self.comment = None
        self.deflateOptions = None
        if opt_params:
            if opt_params.get("flags"):
                self.flags = opt_params["flags"]
            if isinstance(opt_params.get("filename"), str):
                self.filename = opt_params["filename"]
            if isinstance(opt_params.get("comment"), str):
                self.comment = opt_params["comment"]
            if opt_params.get("deflateOptions"):
                self.deflateOptions = opt_params["deflateOptions"]
        if not self.deflateOptions:
            self.deflateOptions = {}
        self.mtime = None  # set to pin the header (the reference reads Date.now(), :77)

    @staticmethod
    def _chars(s):
        out = bytearray()
        for ch in s:
            c = ord(ch)
            if c > 0xFF:
                out.append((c >> 8) & 0xFF)
            out.append(c & 0xFF)
        out.append(0)
        return out

    def compress(self):
        hdr = bytearray([0x1F, 0x8B, 8])
        flg = 0
        if self.flags.get("fname"):
            flg |= Gzip.FlagsMask.FNAME
        if self.flags.get("fcomment"):
            flg |= Gzip.FlagsMask.FCOMMENT
        if self.flags.get("fhcrc"):
            flg |= Gzip.FlagsMask.FHCRC
        hdr.append(flg)
        mtime = int(time.time()) if self.mtime is None else int(self.mtime)
        hdr += bytes([mtime & 0xFF, (mtime >> 8) & 0xFF, (mtime >> 16) & 0xFF, (mtime >> 24) & 0xFF])
        hdr += bytes([0, Gzip.OperatingSystem.UNKNOWN])
        if "fname" in self.flags:  # gated on `!== undefined` (:86)
            hdr += self._chars(self.filename)
        if self.flags.get("comment"):  # the reference gates the bytes on flags.comment (:94)
            hdr += self._chars(self.comment)
        if self.flags.get("fhcrc"):
            crc16 = CRC32.calc(bytes(hdr)) & 0xFFFF
            hdr += bytes([crc16 & 0xFF, (crc16 >> 8) & 0xFF])
        opts = dict(self.deflateOptions)
        buf = np.zeros(max(Gzip.DefaultBufferSize, len(hdr)), dtype=np.uint8)
        buf[: len(hdr)] = np.frombuffer(bytes(hdr), dtype=np.uint8)
        opts["outputBuffer"] = buf
        opts["outputIndex"] = len(hdr)
        rawdeflate = RawDeflate(self.input, opts)  # :107-111
        rawdeflate._want = 1  # CRC-32 of the input (:117) from the same call
        body = rawdeflate.compress()
        crc32 = getattr(rawdeflate, "_crc32", None)
        if crc32 is None:
            crc32 = CRC32.calc(self.input)  # :117
        il = self.input.size & 0xFFFFFFFF
        tr = np.array([crc32 & 0xFF, (crc32 >> 8) & 0xFF, (crc32 >> 16) & 0xFF, (crc32 >> 24) & 0xFF,
                       il & 0xFF, (il >> 8) & 0xFF, (il >> 16) & 0xFF, (il >> 24) & 0xFF], dtype=np.uint8)
        self.output = np.concatenate([body, tr])
        return self.output


class Zip:
    """PKZIP writer, src/zip.js:3-319.  compress() hands all not-yet-compressed
    files to the GPU in one batch (CRC-32 + DEFLATE), then frames them exactly
    as the reference does (:151-273)."""

    class CompressionMethod:
        STORE, DEFLATE = 0, 8

    class OperatingSystem:
        MSDOS, UNIX, MACINTOSH = 0, 3, 7

    class Flags:
        ENCRYPT, DESCRIPTOR, UTF8 = 0x0001, 0x0008, 0x0800

    FileHeaderSignature = [0x50, 0x4B, 0x01, 0x02]
    LocalFileHeaderSignature = [0x50, 0x4B, 0x03, 0x04]
    CentralDirectorySignature = [0x50, 0x4B, 0x05, 0x06]

    def __init__(self, opt_params=None):
        opt_params = opt_params or {}
        self.files = []
        self.comment = opt_params.get("comment")
        self.password = None

    def addFile(self, input, opt_params=None):
        opt_params = opt_params if opt_params is not None else {}
        buf = _as_u8(input)
        compressed, crc32 = False, 0
        if not isinstance(opt_params.get("compressionMethod"), int):
            opt_params["compressionMethod"] = Zip.CompressionMethod.DEFLATE
        if opt_params.get("compress"):
            m = opt_params["compressionMethod"]
            if m == Zip.CompressionMethod.STORE:
                pass
            elif m == Zip.CompressionMethod.DEFLATE:
                crc32 = CRC32.calc(buf)
                buf = self.deflateWithOption(buf, opt_params)
                compressed = True
            else:
                raise ValueError("unknown compression method:%s" % m)
        self.files.append({"buffer": buf, "option": opt_params, "compressed": compressed, "encrypted": False,
                           "size": _as_u8(input).size, "crc32": crc32})

    def setPassword(self, password):
        self.password = password

    def deflateWithOption(self, input, opt_params):
        return RawDeflate(input, opt_params.get("deflateOption")).compress()

    # ---- ZipCrypto (src/zip.js:287-318): key[0..2] are 32-bit words ----
    random = staticmethod(_random.random)  # the reference draws the header bytes from Math.random()

    @staticmethod
    def getByte(key):
        tmp = (key[2] & 0xFFFF) | 2
        return ((tmp * (tmp ^ 1)) >> 8) & 0xFF

    def encode(self, key, n):
        tmp = self.getByte(key)
        self.updateKeys(key, n)
        return tmp ^ n

    @staticmethod
    def updateKeys(key, n):
        key[0] = CRC32.single(key[0], n)
        key[1] = ((key[1] + (key[0] & 0xFF)) * 134775813 + 1) & 0xFFFFFFFF  # 20173 * 6681, :307
        key[2] = CRC32.single(key[2], key[1] >> 24)

    def createEncryptionKey(self, password):
        key = [305419896, 591751049, 878082192]
        for b in bytes(bytearray(int(x) & 0xFF for x in password)):
            self.updateKeys(key, b)
        return key

    def _encode_many(self, key, data):
        """encode() over a buffer with the key words in locals (the cipher is byte-serial: every
        byte's key depends on the plaintext before it)"""
        CRC32.single(0, 0)  # builds the table
        t = CRC32._table
        k0, k1, k2 = key
        out = bytearray(data.size)
        for j, n in enumerate(data.tobytes()):
            tmp = (k2 & 0xFFFF) | 2
            out[j] = (((tmp * (tmp ^ 1)) >> 8) & 0xFF) ^ n
            k0 = t[(k0 ^ n) & 0xFF] ^ (k0 >> 8)
            k1 = ((k1 + (k0 & 0xFF)) * 134775813 + 1) & 0xFFFFFFFF
            k2 = t[(k2 ^ (k1 >> 24)) & 0xFF] ^ (k2 >> 8)
        key[0], key[1], key[2] = k0, k1, k2
        return np.frombuffer(bytes(out), dtype=np.uint8)

    def _compress_pending(self):
        """The batched form of the reference's first loop (:119-137)."""
        pending = [f for f in self.files if not f["compressed"]]
        if not pending:
            return
        # CRC-32 of every pending file (STORE and DEFLATE) + DEFLATE per option group
        groups = {}
        for f in pending:
            m = f["option"]["compressionMethod"]
            if m not in (Zip.CompressionMethod.STORE, Zip.CompressionMethod.DEFLATE):
                raise ValueError("unknown compression method:%s" % m)
            do = f["option"].get("deflateOption") or {}
            ct = do.get("compressionType", 2) if m == Zip.CompressionMethod.DEFLATE else 0
            if ct not in (0, 1, 2):
                raise ValueError("invalid compression type")
            groups.setdefault((m, ct, do.get("lazy", 0) or 0), []).append(f)
        for (m, ct, lazy), fs in groups.items():
            outs, crcs, _ = N.deflate_batch([f["buffer"] for f in fs], ct, lazy, N.F_FINAL_EACH, want_crc=True)
            for f, o, c in zip(fs, outs, crcs):
                f["crc32"] = c
                if m == Zip.CompressionMethod.DEFLATE:
                    f["buffer"] = np.frombuffer(o, dtype=np.uint8)
                    f["compressed"] = True

    def compress(self):
        self._compress_pending()
        files = self.files
        local_size = central_size = 0
        for i, f in enumerate(files):
            opt = f["option"]
            if opt.get("password") is not None or self.password is not None:
                # traditional PKZIP encryption, host arithmetic as in the reference (src/zip.js:138-147):
                # twelve header bytes, then the data, through a byte-serial stream cipher.  The reference
                # fills the header from Math.random() and tests the FILE index against 11 where the
                # format puts a CRC byte into header byte 11; mirrored as it is (its own Unzip does not
                # look at that byte).
                pw = opt.get("password")
                key = self.createEncryptionKey(self.password if pw is None else pw)  # (a numpy array has no truth value)
                plain = f["buffer"]
                enc = np.empty(plain.size + 12, dtype=np.uint8)
                for j in range(12):
                    enc[j] = self.encode(key, (f["crc32"] & 0xFF) if i == 11 else int(self.random() * 256))
                enc[12:] = self._encode_many(key, plain)
                f["buffer"] = enc
            fl = len(opt["filename"]) if opt.get("filename") else 0
            cl = len(opt["comment"]) if opt.get("comment") else 0
            xl = len(opt["extraField"]) if opt.get("extraField") else 0
            local_size += 30 + fl + xl + f["buffer"].size
            central_size += 46 + fl + xl + cl
        eocd = 22 + (len(self.comment) if self.comment else 0)
        out = np.zeros(local_size + central_size + eocd, dtype=np.uint8)
        op1, op2 = 0, local_size
        op3 = op2 + central_size

        def put(pos, vals):
            out[pos:pos + len(vals)] = vals
            return pos + len(vals)

        for f in files:
            opt = f["option"]
            name = bytes(opt["filename"]) if opt.get("filename") else b""
            comment = bytes(opt["comment"]) if opt.get("comment") else b""
            extra = bytes(opt["extraField"]) if opt.get("extraField") else b""  # :231-236
            offset = op1
            op1 = put(op1, Zip.LocalFileHeaderSignature)
            op2 = put(op2, Zip.FileHeaderSignature)
            need = 20
            op2 = put(op2, [need & 0xFF, opt.get("os") or Zip.OperatingSystem.MSDOS])
            date = opt.get("date") or time.localtime()
            if isinstance(date, (tuple, list)):  # (Y, M(1-12), D, h, m, s)
                Y, Mo, D, h, mi, s = date[:6]
            else:
                Y, Mo, D, h, mi, s = date.tm_year, date.tm_mon, date.tm_mday, date.tm_hour, date.tm_min, date.tm_sec
            crc, size, plain = f["crc32"], f["buffer"].size, f["size"]
            method = opt["compressionMethod"]
            pw = opt.get("password")
            pw = self.password if pw is None else pw
            flags = Zip.Flags.ENCRYPT if (pw is not None and len(pw) > 0) else 0  # :177-180
            common = [need & 0xFF, (need >> 8) & 0xFF, flags & 0xFF, (flags >> 8) & 0xFF, method & 0xFF, (method >> 8) & 0xFF,
                      (((mi & 0x7) << 5) | (s // 2)) & 0xFF, ((h << 3) | (mi >> 3)) & 0xFF,
                      ((((Mo) & 0x7) << 5) | D) & 0xFF, ((((Y - 1980) & 0x7F) << 1) | (Mo >> 3)) & 0xFF,
                      crc & 0xFF, (crc >> 8) & 0xFF, (crc >> 16) & 0xFF, (crc >> 24) & 0xFF,
                      size & 0xFF, (size >> 8) & 0xFF, (size >> 16) & 0xFF, (size >> 24) & 0xFF,
                      plain & 0xFF, (plain >> 8) & 0xFF, (plain >> 16) & 0xFF, (plain >> 24) & 0xFF,
                      len(name) & 0xFF, (len(name) >> 8) & 0xFF, len(extra) & 0xFF, (len(extra) >> 8) & 0xFF]
            op1 = put(op1, common)
            op2 = put(op2, common)
            op2 = put(op2, [len(comment) & 0xFF, (len(comment) >> 8) & 0xFF, 0, 0, 0, 0, 0, 0, 0, 0,
                            offset & 0xFF, (offset >> 8) & 0xFF, (offset >> 16) & 0xFF, (offset >> 24) & 0xFF])
            if name:
                op1 = put(op1, np.frombuffer(name, dtype=np.uint8))
                op2 = put(op2, np.frombuffer(name, dtype=np.uint8))
            if extra:
                op1 = put(op1, np.frombuffer(extra, dtype=np.uint8))
                op2 = put(op2, np.frombuffer(extra, dtype=np.uint8))
            if comment:
                op2 = put(op2, np.frombuffer(comment, dtype=np.uint8))
            op1 = put(op1, f["buffer"])
        n = len(files)
        op3 = put(op3, Zip.CentralDirectorySignature)
        op3 = put(op3, [0, 0, 0, 0, n & 0xFF, (n >> 8) & 0xFF, n & 0xFF, (n >> 8) & 0xFF,
                        central_size & 0xFF, (central_size >> 8) & 0xFF, (central_size >> 16) & 0xFF,
                        (central_size >> 24) & 0xFF, local_size & 0xFF, (local_size >> 8) & 0xFF,
                        (local_size >> 16) & 0xFF, (local_size >> 24) & 0xFF])
        cl = len(self.comment) if self.comment else 0
        op3 = put(op3, [cl & 0xFF, (cl >> 8) & 0xFF])
        if self.comment:
            op3 = put(op3, np.frombuffer(bytes(self.comment), dtype=np.uint8))
        return out
