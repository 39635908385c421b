#!/usr/bin/env python3
"""Static VALU instruction mix of the extraction kernels, priced with the per-opcode issue costs measured by
tools/ubench/op_cost.hip on MI355X (gpurun_out/op_cost.txt -> profiles/r02_op_cost.txt): on gfx950 a wave64 VALU
instruction issues in ~2.5 cycles per SIMD only for a small set of opcodes (add / sub / and / or / xor / not / lshr / ashr /
mov, the 16-bit VOP2 forms, f32 add / mul) and in ~4.3 cycles for everything else (min / max / min3 / max3, perm, alignbyte,
packed-16, mul / mad, dot, cmp, cndmask, bcnt / mbcnt, bfe, lshl, any SDWA / DPP form, any form with an SGPR source).
v_fma_f32 / v_fmac_f32 are full rate on normal operands (tools/ubench/fma_forms.hip, profiles/r03_fma_forms.txt; round 2 had timed
them on denormals).  The cycle figures are priced at the idle-chip clock (2.4 GHz): in cycles of the loaded clock (~1.9-2.0 GHz) the two
classes are ~2.1 and ~3.6, i.e. the guide's 2-cycle wave64 issue and its half-rate class; as TIMES per instruction they are what they are.
Compiles orbx_extract.hip / orbx_stereo.hip to ISA (no GPU needed) and prints, per kernel, the instruction counts per class
and the mix-weighted cycles per VALU instruction that bench.py's roofline.issue uses (written into
profiles/r02_sq_counters.json by tools/collect_sq.py)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST_CYC, SLOW_CYC = 2.5, 4.3
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_add_u16", "v_sub_u16", "v_subrev_u16", "v_min_u16", "v_max_u16", "v_min_i16",
        "v_max_i16", "v_mul_lo_u16", "v_lshlrev_b16", "v_lshrrev_b16", "v_ashrrev_i16", "v_add_f16", "v_sub_f16", "v_mul_f16", "v_max_f16",
        "v_min_f16"}


def classify(line):
    t = line.split()
    op = t[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith(("_sdwa", "_dpp")) or "row_" in line or "quad_perm" in line:
        return "slow"
    if base in FAST:
        # an SGPR or VCC source operand drops the instruction to the slow class (k_and_s in op_cost.hip); literals do not
        srcs = " ".join(t[2:])
        if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc", srcs):
            return "slow"
        return "fast"
    return "slow"


def kernel_bodies(asm):
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and line.startswith(".Lfunc_end"):
            out[name] = body; name = None
            continue
        if name:
            s = line.strip()
            if s and not s.startswith((";", ".")):
                body.append(s)
    return out


def pretty(mangled):
    m = re.match(r"_Z(\d+)", mangled)     # Itanium mangling: the name's length precedes it
    base = mangled[m.end():m.end() + int(m.group(1))] if m else mangled
    t = re.search(r"ILi(\d+)ELi(\d+)EE", mangled)
    return base + (f"<{t.group(1)},{t.group(2)}>" if t else "")


def main():
    res = {}
    for src in ("orbx_extract.hip", "orbx_stereo.hip"):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only",
                                   "-o", out, os.path.join(ROOT, "orb-slam2_amd", "csrc", src)], stderr=subprocess.DEVNULL)
            for name, body in kernel_bodies(open(out).read()).items():
                valu = [l for l in body if l.startswith("v_")]
                if not valu:
                    continue
                fast = sum(classify(l) == "fast" for l in valu)
                slow = len(valu) - fast
                res[pretty(name)] = {"valu_static": len(valu), "fast_class": fast, "slow_class": slow,
                                     "salu_static": sum(l.startswith("s_") for l in body), "lds_static": sum(l.startswith("ds_") for l in body),
                                     "cycles_per_valu_inst": round((fast * FAST_CYC + slow * SLOW_CYC) / len(valu), 2)}
    json.dump({"fast_cycles": FAST_CYC, "slow_cycles": SLOW_CYC, "kernels": res}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
