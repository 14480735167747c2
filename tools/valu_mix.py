#!/usr/bin/env python3
"""VALU issue roof of k_extend for ITS instruction mix. tools/valurate measured which vector instructions issue at the full
rate on gfx950 (one wave64 instruction per 2 cycles per SIMD: fp32 fma/mul/add/sub, v_and/v_or/v_xor/v_bitop3, integer
add/sub, v_mov) and which at half rate (4 cycles: everything else; reciprocals / square roots / exp / log: 8). This script
compiles kernels.hip to assembly, takes the shipped instantiation's hot basic blocks (the three written-out node steps
and the triangle step) and prints their cycles per instruction; weighted by the node / triangle wave-steps of the
shipped schedule (bench.py's roofline.shipped_schedule) that is the kernel's mix roof.
usage: python tools/valu_mix.py [node_wave_steps_per_ray tri_wave_steps_per_ray]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = re.compile(r"^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|mac_f32|and_b32|or_b32|xor_b32|bitop3_b32|add_u32|sub_u32|subrev_u32|add_co_u32|mov_b32|pk_mul_f32|pk_fma_f32|pk_add_f32|not_b32)")
SLOW = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos|div_scale|div_fmas|div_fixup)")
def cycles(op):
    if SLOW.match(op): return 8.0 if re.match(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)", op) else 4.0
    return 2.0 if FULL.match(op) else 4.0
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-w",
                           "--cuda-device-only", "-S", "-o", out, "-x", "hip", os.path.join(ROOT, "pathtracer_amd/csrc/kernels.hip")],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2pt8k_extendILb0ELb1ELb0E"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], ["entry", 0, 0.0, 0, 0, 0]
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur); cur = [m.group(1), 0, 0.0, 0, 0, 0]; continue
    if not t or t[0] in ";." : continue
    op = t.split()[0]
    if op.startswith("global_load"): cur[5] += 1
    if op.startswith("v_"):
        cur[1] += 1; cur[2] += cycles(op)
        if op.startswith("v_fma_mix"): cur[3] += 1
        if op.startswith("v_div_") or op.startswith("v_rcp"): cur[4] += 1
blocks.append(cur)
node = [b for b in blocks if b[3] >= 40 and b[1] < 220]          # fast-form node steps (48 v_fma_mix each)
# a triangle test = the block that loads the 48-byte record and shears / evaluates the edge functions + the block with the
# IEEE division (t, u, v); two copies (second record of a group)
tri = [b for b in blocks if (b[5] == 3 and 50 <= b[1] <= 90) or (b[4] == 5 and 25 <= b[1] <= 45 and b[5] == 0 and b[3] == 0)][:4]
def summ(name, bs, per=1):
    n, c = sum(b[1] for b in bs), sum(b[2] for b in bs)
    k = max(1, len(bs) // per)
    print(f"{name}: {k} copies, {n / k:.0f} VALU instructions each, {c / max(1, n):.2f} cycles per instruction")
    return n / k, c / k
ni, nc = summ("node step (fast form)", node)
ti, tc = summ("triangle test", tri, 2)
alln, allc = sum(b[1] for b in blocks), sum(b[2] for b in blocks)
print(f"whole kernel, static: {alln} VALU instructions, {allc / alln:.2f} cycles per instruction")
if len(sys.argv) > 2:
    wn, wt = float(sys.argv[1]), float(sys.argv[2])
    cpi = (wn * nc + wt * tc) / (wn * ni + wt * ti)
    print(f"weighted by {wn} node and {wt} triangle wave-steps per ray: {cpi:.2f} cycles per instruction -> mix roof "
          f"{256 * 4 * 2.4 / cpi:.0f} G wave-instructions/s at 2.4 GHz ({256 * 4 * 2.4 / 2:.1f} for an all-full-rate kernel)")
