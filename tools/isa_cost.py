"""Static instruction mix of one kernel of an AMDGPU .s file, weighted by the issue classes tools/valu_probe.hip measured on gfx950
(profiles/archive/r02_valu_probe.txt): full rate = 1 (v_fma/add/mul/sub_f32, v_mov, v_add_u32, and/or/xor), half rate = 1.8 (conversions,
min/max, compares, selects, shifts, 3-operand integer, f64, packed f32, anything with an SGPR source operand — literal and inline
constants cost nothing), transcendental = 3.5.

usage: python tools/isa_cost.py file.s kernel_substring [--loop LABEL]   (straight-line kernels: static ~ dynamic)
"""
import re
import sys
from collections import Counter

FULL = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mac_f32", "v_nop", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def analyse(path, name):
    """(valu_instructions, weighted_cost, classes, sgpr_slowed, sections) of the kernel whose mangled name contains `name`."""
    return _walk(path, name)[:5]


def main():
    path, name = sys.argv[1], sys.argv[2]
    n_valu, cost, cls, sgpr_src, sections, ops, n_lines = _walk(path, name)
    sect_n, sect_cost = sections
    print(f"{name}: lines {n_lines}, weighted VALU cost {cost:.0f}, classes {dict(cls)}, full-rate ops slowed by an SGPR/literal source {sgpr_src}")
    print("  sections (VALU count / weighted):", ", ".join(f"{k} {sect_n[k]}/{sect_cost[k]:.0f}" for k in sect_n))
    print("  waitcnt", ops["s_waitcnt"], " branches", sum(v for k, v in ops.items() if k.startswith("s_cbranch")), " s_nop", ops["s_nop"])
    for k, v in ops.most_common(45):
        print(f"  {v:5d} {k}")


def _walk(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = Counter()
    cost = 0.0
    cls = Counter()
    sgpr_src = 0
    sect, sect_cost, sect_n = "entry", Counter(), Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"; MARK (\w+)", t)
        if m:
            sect = m.group(1)
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
        ops[base] += 1
        if op.startswith("v_") and not op.startswith("v_readfirstlane") and not op.startswith("v_readlane"):
            args = t[len(op):]
            srcs = args.split(",")[1:] if "," in args else []
            has_s = any(re.search(r"\bs\d+|\bs\[|\bvcc\b", a) for a in srcs) and not base.startswith(("v_cmp", "v_cndmask"))     # literals and inline constants are free (probe)
            if base in TRANS:
                c = 3.5; cls["trans"] += 1
            elif base in FULL and not has_s:
                c = 1.0; cls["full"] += 1
            else:
                c = 1.8; cls["half"] += 1
                if base in FULL:
                    sgpr_src += 1
            cost += c
            sect_cost[sect] += c; sect_n[sect] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            cls[op.split("_")[0] + ("_load" if "load" in op else "_store" if "store" in op else "_atomic")] += 1
        elif op.startswith("ds_"):
            cls["lds"] += 1
    return sum(sect_n.values()), cost, cls, sgpr_src, (sect_n, sect_cost), ops, end - start


if __name__ == "__main__":
    main()
