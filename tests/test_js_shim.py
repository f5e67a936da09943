"""The JavaScript host side: N-API addon + shim with the reference's class API.

CPU part: the addon builds, loads under Node and exposes its functions, and the
host-only arithmetic (checksum combine) works without a GPU.  GPU part: the
reference's compression test cases, restated in zlib.es_amd/js/test/run_tests.js,
run under Node against the GPU-backed shim.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "zlib.es_amd", "js")
NODE = shutil.which("node")

needs_node = pytest.mark.skipif(NODE is None, reason="node not installed")


def _node(script, timeout=120):
    return subprocess.run([NODE, "-e", script], cwd=JS, capture_output=True, text=True, timeout=timeout)


@needs_node
def test_addon_loads_and_exports():
    if not os.path.exists(os.path.join(JS, "zgpu_addon.node")):
        pytest.fail("zgpu_addon.node missing: run __graft_entry__.build() / make -C zlib.es_amd/js")
    r = _node("const a=require('./zgpu_addon.node'); console.log(Object.keys(a).sort().join(','))")
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "adler32Combine,allocInput,checksum,crc32Combine,deflateBatch,deflateBound,hostDevices,hostTrim,init,setDenseSearch,setLongMatch,setMinLazy,setSearch,setSplit"


@needs_node
def test_shim_surface_and_host_arithmetic():
    r = _node("""
const z = require('./zlib_gpu.js');
const names = ['RawDeflate','Deflate','Gzip','Zip','CRC32','Adler32'];
for (const n of names) if (typeof z.Zlib[n] !== 'function') throw new Error('missing ' + n);
for (const m of ['makeNocompressBlock','makeFixedHuffmanBlock','makeDynamicHuffmanBlock','compress'])
  if (typeof z.RawDeflate.prototype[m] !== 'function') throw new Error('missing method ' + m);
if (z.Deflate.CompressionType.DYNAMIC !== 2 || z.RawDeflate.CompressionType.NONE !== 0) throw new Error('enum');
// host arithmetic through the addon (no GPU needed): crc32("123456789") from its two halves
const c = z.addon.crc32Combine(0x9be3e0a3 /* crc32("1234") */, 0x131da070 /* crc32("56789") */, 5);
if ((c >>> 0) !== 0xcbf43926) throw new Error('combine ' + (c>>>0).toString(16));
if (z.CRC32.single(0x12, 0x34) !== ((z.CRC32.Table[(0x12 ^ 0x34) & 0xff] ^ 0) >>> 0)) throw new Error('single');
console.log('ok');
""")
    assert r.returncode == 0, r.stderr + r.stdout
    assert "ok" in r.stdout


@needs_node
def test_no_cpu_fallback_in_js():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _node("const z=require('./zlib_gpu.js'); try { z.CRC32.calc(new Uint8Array(10)); console.log('computed'); } "
              "catch (e) { console.log('threw: ' + e.message); }")
    assert r.returncode == 0, r.stderr
    assert "threw" in r.stdout and "computed" not in r.stdout


@pytest.mark.gpu
@needs_node
def test_named_devices_under_node():
    """ZGPU_DEVICES names the GPUs of the library's batch split.  The one GPU of the box is named twice, so the
    split over two devices runs for real under Node; the streams must be correct."""
    env = dict(os.environ, ZGPU_DEVICES="0,0")
    r = subprocess.run([NODE, "-e", """
const z = require('./zlib_gpu.js'), zlib = require('zlib');
const devs = z.addon.hostDevices();
if (!Array.isArray(devs) || devs.length !== 2) throw new Error('devices ' + JSON.stringify(devs));
const n = 300, ub = 1 << 20, big = new Uint8Array(n * ub);            // 300 MiB: above the multi-device threshold
let x = 7; for (let i = 0; i < big.length; ++i) { if ((i & 3) === 0) x = (Math.imul(x, 1664525) + 1013904223) >>> 0; big[i] = 97 + ((x >>> (8 * (i & 3))) & 7); }
const views = []; for (let i = 0; i < n; ++i) views.push(big.subarray(i * ub, (i + 1) * ub));
const outs = z.Zlib.RawDeflate.compressBatch(views, { compressionType: 2 });
for (const i of [0, 1, 149, 150, 298, 299]) {
  const back = zlib.inflateRawSync(Buffer.from(outs[i].buffer, outs[i].byteOffset, outs[i].length));
  if (Buffer.compare(back, Buffer.from(big.buffer, i * ub, ub)) !== 0) throw new Error('unit ' + i);
}
console.log('ok ' + devs.join(','));
"""], cwd=JS, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-2000:] + "\n" + r.stderr[-2000:])
    assert r.stdout.startswith("ok ")


@pytest.mark.gpu
@needs_node
def test_reference_cases_under_node():
    r = subprocess.run([NODE, os.path.join(JS, "test", "run_tests.js")], cwd=JS, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert "tests passed" in r.stdout
