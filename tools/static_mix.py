"""Static instruction mix of the main loop of a blocked TV-L1 kernel (k_iterate_tbr<T, PPL, PZ, WPS, PF, MODE>), per pipeline
stage and pixel row: compiles tvl1_tbr_kernels.hip to gfx950 assembly, finds the largest loop of the instantiation and counts its
instructions by class.  bench.py's `roofline` (bound "valu_issue") uses these figures; the output of record is
profiles/static_mix_tbr.json (the default kernel: joined waves, barrier form) and profiles/static_mix_tbr_jw0.json (independent waves).
Usage: python tools/static_mix.py [T PPL PZ WPS PF MODE JW [NG P16]]  (default 10 1 0 4 2 0 0; NG defaults to what a default calc runs)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mix(T=10, PPL=1, PZ=0, WPS=4, PF=2, MODE=0, JW=0, NG=None, P16=0, FW=0, GAM=0):
    # NG: the instantiation without a |grad|^2 plane -- what a default calc runs since round 4 for T = 10, JW = 2 (tb_nograd_ok)
    if NG is None:
        NG = 1 if (T == 10 and JW == 2 and MODE == 0) else 0
    src = os.path.join(ROOT, "opencv_contrib_amd", "csrc", "tvl1_tbr_kernels.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-x", "hip", "-S", "--cuda-device-only", src,
                        "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    pat = f"k_iterate_tbrILi{T}ELi{PPL}ELb{PZ}ELi{WPS}ELi{PF}ELi{MODE}ELi{JW}ELb{NG}ELb{P16}ELi{FW}ELb{GAM}EE"   # FW (round 5): 0 = the warp as its own launch; GAM (round 6): 1 = with the illumination channel
    for f in re.split(r"\n\s*\.globl\s+", txt):
        if pat not in f.split("\n", 1)[0]:
            continue
        lines = f.split("\n")
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
                best = (labels[m.group(1)], i)
        cnt = collections.Counter()
        for l in lines[best[0]:best[1] + 1]:
            l = l.strip()
            if l and not l.startswith((".", ";", "//")) and not l.endswith(":"):
                cnt[l.split()[0]] += 1
        cls = collections.Counter()
        for op, n in cnt.items():
            if op.startswith("v_"):
                if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)", op):
                    cls["transcendental"] += n
                elif "dpp" in op:
                    cls["dpp"] += n
                elif op.startswith("v_cndmask"):
                    cls["cndmask"] += n
                else:
                    cls["valu_plain"] += n
            elif op.startswith("s_"):
                cls["salu"] += n
            elif op.startswith("ds_"):
                cls["lds"] += n
            elif op.startswith(("global_", "buffer_")):
                cls["vmem"] += n
            else:
                cls["other"] += n
        stages = (T + 1 + PF) * T * PPL   # the loop body is unrolled P = T + 1 + PF steps of T stages
        rw = rate_weighted_cycles(cnt)
        return {"kernel": pat, "loop_instructions": sum(cnt.values()), "per_stage_and_pixel": {k: round(v / stages, 3) for k, v in cls.items()},
                # SIMD cycles of a stage (one pixel row of one wave) at the MEASURED per-operation issue rates (profiles/valu_rates_gfx950.json)
                "rate_weighted_per_stage": {"full_rate": round(rw["full_rate"] / stages, 3), "half_rate": round(rw["half_rate"] / stages, 3),
                                            "transcendental": round(rw["transcendental"] / stages, 3), "cycles": round(rw["cycles"] / stages, 2)}}
    raise SystemExit("instantiation not found: " + pat)


def rate_weighted_cycles(cnt):
    """SIMD cycles of one trip of a loop if every VALU instruction issued at its measured gfx950 rate (profiles/valu_rates_gfx950.json): the
    2-cycle 'spec' rate holds for a short list of operations only; SDWA, DPP, selects, min / max, 24-bit multiplies, left shifts, ...
    take twice as long."""
    r = json.load(open(os.path.join(ROOT, "profiles", "valu_rates_gfx950.json")))
    full = set(r["full_rate_ops"])
    c = r["cycles_per_wave64_instruction"]
    nf = nh = nt = 0
    for op, n in cnt.items():
        if not op.startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)", op):
            nt += n
        elif base in full and not op.endswith(("_sdwa", "_dpp")):
            nf += n
        else:
            nh += n
    return {"full_rate": nf, "half_rate": nh, "transcendental": nt, "cycles": nf * c["full_rate"] + nh * c["half_rate"] + nt * c["transcendental"]}


def mix_sbm(R=7, MODE=0, WT=None):
    """Static instruction count of the row loop of k_block_match<R, MODE> (stereobm_kernels.hip): per trip a lane (= one disparity)
    slides the column sums of its NC = TW + 2R tile columns down one row and produces the TW window SSDs / winners of that row, so
    loop VALU / TW = lane-instructions per (output pixel, disparity) of a tile row.  Output of record: profiles/static_mix_sbm.json."""
    src = os.path.join(ROOT, "opencv_contrib_amd", "csrc", "stereobm_kernels.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                        "-DMI_STATIC_MIX_INTERIOR_TILES",   # the edge-emulation variant of the row stage (right-most tiles only) is compiled out
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-x", "hip", "-S", "--cuda-device-only", src,
                        "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    name = f"_ZN2mi3sbm13k_block_matchILi{R}ELi{MODE}ELb{1 if WT else 0}EEEvNS0_6BmArgsE"
    i = txt.index("\n" + name + ":")
    lines = txt[i:txt.index(".Lfunc_end", i)].split("\n")
    labels = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for k, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), k) < k and (best is None or k - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], k)
    def count(a, b):
        c = collections.Counter()
        for l in lines[a:b + 1]:
            l = l.strip()
            if l and not l.startswith((".", ";", "//")) and not l.endswith(":"):
                c[l.split()[0]] += 1
        return c
    cnt = count(*best)
    # the first 2R rows of a band only add their row to the column sums and `continue`: the shortest backward branch that still
    # contains the byte-extracting subtract
    warm = None
    for k, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), k) < k:
            c = count(labels[m.group(1)], k)
            if c.get("v_sub_u32_sdwa", 0) > 0 and (warm is None or sum(c.values()) < sum(warm.values())):
                warm = c
    tw = 48 if R <= 12 else max(16, (64 - 2 * R) & ~3)
    valu = sum(n for o, n in cnt.items() if o.startswith("v_"))
    rw = rate_weighted_cycles(cnt)
    return {"kernel": f"k_block_match<{R},{MODE},{'true' if WT else 'false'}>", "row_loop_instructions": sum(cnt.values()), "row_loop_valu": valu,
            "rate_weighted": rw, "rate_weighted_cycles_per_output_pixel_and_disparity_lane": round(rw["cycles"] / tw, 3),
            "warmup_row_valu": sum(n for o, n in (warm or {}).items() if o.startswith("v_")),
            "row_loop_salu": sum(n for o, n in cnt.items() if o.startswith("s_")), "row_loop_lds": sum(n for o, n in cnt.items() if o.startswith("ds_")),
            "tile_output_columns": tw, "valu_per_output_pixel_and_disparity": round(valu / tw, 3),
            "top": dict(cnt.most_common(10))}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sbm":   # python tools/static_mix.py sbm [R MODE]
        print(json.dumps(mix_sbm(*[int(x) for x in sys.argv[2:5]]), indent=1))   # R MODE WT
    else:
        a = [int(x) for x in sys.argv[1:10]] or [10, 1, 0, 4, 2, 0, 0]
        print(json.dumps(mix(*a), indent=1))
