#!/usr/bin/env python3
"""Counted instruction budget of the encoder kernel's frame loop, from the ISA hipcc emits for gfx950 (no GPU needed).
Compiles csrc/ita_plugin.hip to assembly, takes ita_stream_kernel<64, true, 1> from its frame-loop header to the end of the
function (the loop body; inline-asm requantisation blocks are expanded in the assembly) and groups the opcodes.
Counts are per WAVE and frame; a frame is 8 waves.  usage: python tools/valu_budget.py [extra hipcc flags]
The single-rounding permission of the requantisation sites is a template argument: the default counts the FAST instantiation
(all six sites of the layer proven at load time), --exact the other one; -DITA_RQ_STYLE=2 is the round-2 form."""
import collections, os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "drone-oa-iree-vit-accelerator_amd", "csrc", "ita_plugin.hip")
# <E = 64, FFN, TOK = 1, no stamps, f32 I/O, FAST>: the instantiation the bench blob runs; --exact counts the one with FAST = false
KERNEL = "_Z17ita_stream_kernelILi64ELb1ELi1ELb0ELb0ELb%dEEv13ItaStreamArgs" % (0 if "--exact" in sys.argv else 1)
CLASSES = [
    ("requantise: unbias + scale + round (v_pk_add/mul/fma_f32, v_pk_mov)", r"^v_pk_(add|mul|fma)_f32|^v_pk_mov"),
    ("requantise: float clamp (v_med3_f32)", r"^v_med3_f32"),
    ("requantise: round + byte insert (SDWA add)", r"^v_add_f32_sdwa"),
    ("requantise: 16-bit pairs, u8 saturation, sign flip (v_perm, v_sat_pk, v_xor)", r"^v_perm_b32|^v_sat_pk_u8_i16|^v_xor_b32"),
    ("integer softmax (packed 16-bit)", r"^v_pk_.*(i16|u16|b16)|^v_dot4|^v_rndne|^v_permlane|^v_cndmask"),
    ("float: LayerNorm, residuals, dequantise, f16 split", r"^v_(add|sub|mul|fma|fmac|rcp|rsq|sqrt|div|cvt_pk_f16|cvt_f16|cvt_f32_f16|max|min)_?"),
    ("tokenizer blend / window (integer)", r"^v_(mul_u32_u24|mad_u32_u24|alignbyte|alignbit|cvt_f32_u32|bfe|and_b32|or_b32|lshl|lshr|add_u32|add3|lshl_add)"),
    ("moves", r"^v_mov|^v_accvgpr|^v_readfirstlane"),
]


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I",
                               os.path.join(REPO, "include"), "--cuda-device-only", "-S", "-o", out, SRC] + [a for a in sys.argv[1:] if a != "--exact"],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    beg = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(beg, len(lines)) if "s_endpgm" in lines[i])
    body = lines[beg:end]
    hdr = next(i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:.*Loop Header", l))
    ops = collections.Counter()
    for l in body[hdr:]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        ops[l.split()[0]] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma")}
    total = sum(valu.values())
    print(f"ita_stream_kernel<64, true, 1> frame loop, per wave and frame: {sum(ops.values())} instructions")
    print(f"  VALU {total}   MFMA i8 {ops.get('v_mfma_i32_16x16x64_i8', 0)}   MFMA f32 {ops.get('v_mfma_f32_16x16x4_f32', 0)}   "
          f"LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}   VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'flat_')))}   "
          f"SALU/other {sum(v for k, v in ops.items() if k.startswith('s_'))} (s_nop {ops.get('s_nop', 0)}, s_waitcnt {ops.get('s_waitcnt', 0)})")
    left = dict(valu)
    print(f"  {'VALU class':58s} {'per wave':>9s} {'per frame':>10s} {'share':>6s}")
    for name, pat in CLASSES:
        n = sum(v for k, v in left.items() if re.match(pat, k))
        left = {k: v for k, v in left.items() if not re.match(pat, k)}
        print(f"  {name:58s} {n:9d} {8 * n:10d} {100.0 * n / total:5.1f}%")
    n = sum(left.values())
    print(f"  {'other (' + ', '.join(sorted(left, key=left.get, reverse=True)[:4]) + ' ...)':58s} {n:9d} {8 * n:10d} {100.0 * n / total:5.1f}%")
    print(f"  {'all':58s} {total:9d} {8 * total:10d}")


if __name__ == "__main__":
    main()
