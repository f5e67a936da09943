"""Traditional PKZIP encryption in the Python mirror of Zip (host arithmetic, no GPU): the key
schedule and the cipher stream against vectors made by the reference's own Zip#createEncryptionKey /
#encode under Node (tests/golden/gen_golden.mjs, src/zip.js:287-318), and against CPython's
zipfile decrypter, an implementation that never saw this one."""
import zipfile

import numpy as np

import _oracle as zo
import zlib_es_amd as z


def test_key_schedule_and_cipher_match_the_reference(golden):
    for v in golden["zipcrypto"]:
        zp = z.Zip()
        key = zp.createEncryptionKey(v["password"])
        assert key == v["key_after_password"]
        plain = zo.gen(v["spec"])
        enc = bytes(zp.encode(key, int(b)) for b in plain)
        assert enc.hex() == v["out"]["hex"] and key == v["key_after"]
        # the buffer form is the same cipher
        key2 = zp.createEncryptionKey(v["password"])
        assert zp._encode_many(key2, plain).tobytes() == enc and key2 == v["key_after"]


def test_cipher_decrypts_with_cpython_zipfile():
    rng = np.random.default_rng(3)
    for n, pw in ((0, b"a"), (1, b"hogefugapiyo"), (5000, bytes(range(10)))):
        plain = rng.integers(0, 256, n, dtype=np.uint8)
        zp = z.Zip()
        key = zp.createEncryptionKey(list(pw))
        head = bytes(zp.encode(key, int(b)) for b in range(12))
        body = zp._encode_many(key, plain).tobytes()
        dec = zipfile._ZipDecrypter(pw)
        assert dec(head) == bytes(range(12)) and dec(body) == plain.tobytes()
