"""CPU gate on the SHIPPED code objects: no kernel of libdiinn_hip.so may use scratch memory (a register spill to
scratch is a silent several-fold slowdown: round 2 shipped liif_kernel with 954 spilled registers and
decode_bf16x2_kernel with 360 while the docs said "no kernel uses scratch"), and every kernel must keep the occupancy its
design counts on (one wave per SIMD for the register-resident kernels, two where two workgroups or
two waves are meant to cover each other).

Reads the AMDGPU metadata notes of the gfx950 code objects embedded in the library (clang offload bundles in
.hip_fatbin), i.e. what will actually be loaded on the GPU box -- not a separate compile."""
import os
import shutil
import struct
import subprocess

import pytest
import yaml

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LDS_PER_CU = 160 * 1024
REGS_PER_SIMD = 512

# kernel name (as in the mangled symbol) -> waves per SIMD the design needs.  Anything not listed needs >= 1.
DESIGN_OCCUPANCY = {
    "decode_bf16_coop8_kernel": 2,       # two waves per SIMD cover each other's vector-memory stalls
    "precompute_P_kernel": 2,            # two workgroups per CU hide each other's waits (4.2)
    "precompute_P_bf16_wide_kernel": 2,
    "conv_wino_half_kernel": 2,          # two workgroups per CU cover prologue / epilogue (4.8)
    "conv_wino4_kernel": 4,              # one workgroup of 16 waves per CU: 12 MFMA waves on three SIMDs, 4 transform waves on the fourth (3.9)
    "conv1x1_stream_kernel": 2,
    "conv_ksplit_kernel_3x3": 2,
    "conv_ksplit_kernel_3x3_lat": 2,
    "conv_ksplit_kernel_1x1": 2,
}


def code_objects(lib_path):
    data = open(lib_path, "rb").read()
    pos, out = 0, []
    while True:
        i = data.find(BUNDLE_MAGIC, pos)
        if i < 0:
            return out
        (num,) = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(num):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and s > 0:
                out.append(data[i + o:i + o + s])
        pos = i + len(BUNDLE_MAGIC)


def kernel_metadata(lib_path, tmp_path):
    kernels = []
    for n, blob in enumerate(code_objects(lib_path)):
        f = tmp_path / f"co{n}.elf"
        f.write_bytes(blob)
        txt = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
        start = txt.index("---") + 3
        end = txt.index("\n...", start) if "\n..." in txt[start:] else len(txt)
        meta = yaml.safe_load(txt[start:end])
        kernels += meta["amdhsa.kernels"]
    return kernels


def base_name(mangled):
    import re
    m = re.match(r"_Z(\d+)", mangled)
    n = int(m.group(1))
    return mangled[m.end():m.end() + n]


def occupancy(k):
    regs = -(-int(k[".vgpr_count"]) // 8) * 8                     # unified VGPR+AGPR file, allocated in blocks of 8
    waves_regs = min(8, REGS_PER_SIMD // max(regs, 8))
    lds = int(k[".group_segment_fixed_size"])
    waves_per_wg = -(-int(k[".max_flat_workgroup_size"]) // 64)
    if lds:
        waves_lds = (LDS_PER_CU // lds) * waves_per_wg / 4.0
    else:
        waves_lds = 8
    return min(waves_regs, waves_lds)


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf (ROCm) not installed")
def test_no_shipped_kernel_uses_scratch_and_occupancy_is_as_designed(tmp_path):
    import diinn_amd._native as N
    ks = kernel_metadata(N.LIB_PATH, tmp_path)
    names = {base_name(k[".name"]) for k in ks}
    # the hot-path kernels are all in the library that was inspected
    for must in ("decode_kernel", "decode_coop16_kernel", "precompute_P_wino_kernel", "precompute_P_kernel", "liif_kernel",
                 "metasr_kernel", "decode_bf16x2_kernel", "decode_bf16_coop8_kernel", "conv_wino_kernel", "conv_wino4_kernel", "bwd_layer_kernel"):
        assert must in names, f"{must} not found in {N.LIB_PATH}"
    assert len(ks) >= 50
    bad = []
    for k in ks:
        name = base_name(k[".name"])
        if int(k[".private_segment_fixed_size"]) != 0 or k.get(".uses_dynamic_stack"):
            bad.append(f"{k['.name']}: {k['.private_segment_fixed_size']} B/lane of scratch "
                       f"({k['.vgpr_spill_count']} VGPRs spilled)")
        # (SGPR spills go to VGPR lanes, not to memory: the training forward and the encoder's Winograd kernels keep a few
        # descriptors there by design; only scratch is gated)
        need = DESIGN_OCCUPANCY.get(name, 1)
        if occupancy(k) < need:
            bad.append(f"{k['.name']}: occupancy {occupancy(k)} waves/SIMD < designed {need} "
                       f"({k['.vgpr_count']} registers, {k['.group_segment_fixed_size']} B LDS)")
    assert not bad, "\n".join(bad)


def test_no_inline_asm_statement_has_an_asynchronous_output():
    """Source gate (round 6): an asm statement whose OUTPUT operand is written by an asynchronous instruction -- a vector / scalar
    memory load or an LDS read -- is a latent corruption: hipcc takes the output for written at the statement and may copy it out
    and reuse the register while the load is in flight (cdna_hip_programming.md 5.7).  The P kernels' L2 warm-up touches were such
    statements for four rounds and went wrong the day an unrelated edit changed the register allocation.  Loads
    belong in builtins the compiler counts in vmcnt / lgkmcnt; the one exception is a statement that waits for its own result
    inside the statement (the s_memtime / s_memrealtime stamps: "... s_waitcnt lgkmcnt(0)")."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = sorted(glob.glob(os.path.join(root, "dual-interactive-implicit-neural-network_amd", "csrc", "*")))
    assert len(srcs) >= 15
    asynchronous = re.compile(r"\b(global_load|buffer_load|flat_load|scratch_load|ds_read|ds_load|s_load|s_buffer_load|s_memtime|s_memrealtime|global_atomic|buffer_atomic)")
    bad = []
    for path in srcs:
        text = open(path).read()
        for m in re.finditer(r"asm\s+volatile\s*\((.*?)\)\s*;", text, flags=re.S):
            stmt = m.group(1)
            parts = stmt.split(":")
            code = parts[0]
            outputs = parts[1].strip() if len(parts) > 1 else ""
            if not outputs or not asynchronous.search(code):
                continue
            if re.search(r"s_mem(real)?time", code) and "s_waitcnt lgkmcnt(0)" in code and not re.search(r"(global|buffer|flat|ds)_", code):
                continue                                         # waits for its own result inside the statement
            bad.append((os.path.basename(path), text[:m.start()].count("\n") + 1, " ".join(code.split())[:80]))
    assert not bad, bad
