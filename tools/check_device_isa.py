#!/usr/bin/env python3
"""Asserts properties of the gfx950 device code inside the BUILT libm3asr_hip.so (not of the sources, not of the flags).

    python tools/check_device_isa.py [path/to/libm3asr_hip.so]

The library is built without packed-FP32 VALU instructions (Makefile NOPK): a toolchain that ignores the
-target-feature switch, or a `make NOPK=`, would silently bring back the ~1 % wrong-result rate of concurrent execution
contexts.  This walks the clang offload bundles of the library's .hip_fatbin section, disassembles every
hipv4-amdgcn-amd-amdhsa--gfx950 code object with llvm-objdump and counts v_pk_{add,mul,fma}_f32.  Used by
__graft_entry__.build(), tests/test_abi.py (CPU tier) and bench.py (the JSON line records the count and `hipcc --version`).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PACKED_F32 = re.compile(rb"\bv_pk_(add|mul|fma)_f32\b")
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3m-asr-inference_amd", "m3asr", "libm3asr_hip.so")


def _fatbin_bytes(lib):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fatbin")
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, out])
        return open(out, "rb").read()


def device_code_objects(lib, arch="gfx950"):
    """[(bundle index, triple, bytes)] for every device code object of `arch` in the library."""
    blob = _fatbin_bytes(lib)
    found, at, idx = [], blob.find(MAGIC), 0
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if triple.startswith("hip") and triple.endswith(arch) and size:
                found.append((idx, triple, blob[at + off:at + off + size]))
        idx += 1
        at = blob.find(MAGIC, at + 1)
    if blob.find(b"CCOB") >= 0 and not found:
        raise RuntimeError("%s holds compressed offload bundles; rebuild without --offload-compress" % lib)
    return found


def scan(lib=DEFAULT_LIB):
    """{'code_objects': n, 'kernels': n, 'instructions': n, 'packed_fp32': n, 'per_object': {...}}"""
    objs = device_code_objects(lib)
    if not objs:
        raise RuntimeError("no gfx950 code object found in %s" % lib)
    res = {"code_objects": len(objs), "kernels": 0, "instructions": 0, "packed_fp32": 0, "per_object": {}}
    with tempfile.TemporaryDirectory() as td:
        for idx, _triple, data in objs:
            path = os.path.join(td, "co%d.hsaco" % idx)
            open(path, "wb").write(data)
            dis = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", path],
                                 stdout=subprocess.PIPE, check=True).stdout
            kernels = len(re.findall(rb"^[0-9a-f]+ <[^>]+>:$", dis, re.M))
            insns = len(re.findall(rb"^\s+[sv]_\w+|^\s+(?:ds|buffer|global|flat|scratch)_\w+", dis, re.M))
            pk = len(PACKED_F32.findall(dis))
            res["kernels"] += kernels
            res["instructions"] += insns
            res["packed_fp32"] += pk
            if pk:
                res["per_object"][idx] = pk
    return res


def hipcc_version():
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode()
        hip = re.search(r"HIP version:\s*(\S+)", out)
        clang = re.search(r"clang version\s*(\S+)", out)
        return "HIP %s / clang %s" % (hip.group(1) if hip else "?", clang.group(1) if clang else "?")
    except Exception as e:          # noqa: BLE001  (a missing compiler on a box is a fact to report, not to die on)
        return "unavailable (%s)" % type(e).__name__


def assert_no_packed_fp32(lib=DEFAULT_LIB):
    r = scan(lib)
    if r["instructions"] < 10000:
        raise AssertionError("disassembly of %s looks empty (%d instructions): the check did not see the device code" % (lib, r["instructions"]))
    if r["packed_fp32"]:
        raise AssertionError("%d v_pk_{add,mul,fma}_f32 instructions in %s (code objects %s): the NOPK build switch did not take "
                             "effect (DESIGN.md 10.8)" % (r["packed_fp32"], lib, sorted(r["per_object"])))
    return r


if __name__ == "__main__":
    r = assert_no_packed_fp32(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB)
    print("ok: %d code objects, %d functions, %d instructions, 0 packed-FP32 VALU; %s" % (r["code_objects"], r["kernels"], r["instructions"], hipcc_version()))
