"""Where does a decode lane's time go?  Reads a rocprofv3 --kernel-trace CSV of the LANES schedule and, per hardware queue,
splits the wall time of the decode phase into kernel time and the gaps between consecutive dispatches of the same queue
(a decode is one dependent chain per queue, so a gap is launch / dependency latency, not idle work).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python3 bench.py --steps 8 --warmup 0 --tokens 30 --no-cpu-baseline
    python3 tools/lane_gap_analysis.py gpurun_out/prof_gap > gpurun_out/lane_gaps.txt
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(cross_attn_rows_kernel|cross_attn_kernel|self_attn_kernel|dec_gemm_kernel|sampler_kernel|embed_kernel|gemm256_kernel|"
                  r"encoder_attention_kernel|layernorm_kernel|mel_\w+_kernel|beam_\w+_kernel)", name)
    base = m.group(1) if m else name[:40]
    if base == "dec_gemm_kernel":
        m2 = re.search(r"Li(\d)ELb([01])ELi(\d)ELi(\d)", name) or re.search(r"int, E(L?), (bool|true|false), (?:E, )?(\d), (\d)", name)
        if m2:
            base += "<" + ",".join(str(g) for g in m2.groups()) + ">"
    return base


def main():
    d = sys.argv[1]
    files = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    byq = defaultdict(list)
    for q, s, e, n in rows:
        byq[q].append((s, e, n))
    print(f"{len(rows)} dispatches on {len(byq)} queues")
    t_first = min(r[1] for r in rows)
    for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        lst.sort()
        dec = [x for x in lst if x[2].startswith(("cross_attn", "self_attn", "dec_gemm", "sampler", "embed"))]
        if len(dec) < 1000:
            print(f"queue {q}: {len(lst)} dispatches, {len(dec)} of the decoder - skipped")
            continue
        # decode segments: consecutive decoder dispatches with no other kernel of this queue in between and gaps < 1 ms
        kt = 0
        segs, long_gaps = [], []
        gaps = defaultdict(list)
        seg_wall = 0
        seg_start = dec[0][0]
        prev = dec[0]
        kt += prev[1] - prev[0]
        kern = defaultdict(list)
        kern[prev[2]].append(prev[1] - prev[0])
        for cur in dec[1:]:
            g = cur[0] - prev[1]
            if g > 1_000_000:                      # another group's front ends ran in between
                seg_wall += prev[1] - seg_start
                segs.append((seg_start, prev[1]))
                long_gaps.append(g)
                seg_start = cur[0]
            else:
                gaps[(prev[2], cur[2])].append(g)
            kt += cur[1] - cur[0]
            kern[cur[2]].append(cur[1] - cur[0])
            prev = cur
        seg_wall += prev[1] - seg_start
        segs.append((seg_start, prev[1]))
        print(f"queue {q}: decode segments (ms, relative to the first dispatch of the trace): " +
              ", ".join(f"{(a - t_first) / 1e6:.1f}..{(b - t_first) / 1e6:.1f}" for a, b in segs) +
              f"; gaps over 1 ms between them: {[round(g / 1e6, 1) for g in long_gaps]}")
        # gaps of 20 us .. 1 ms inside a segment: host round trips (the greedy loop reads n_done every few iterations)
        mid = [g for v in gaps.values() for g in v if g > 20_000]
        print(f"   gaps of 20 us .. 1 ms inside the segments: {len(mid)}, {sum(mid) / 1e6:.1f} ms in total")
        allg = [g for v in gaps.values() for g in v]
        print(f"queue {q}: {len(dec)} decoder dispatches, decode wall {seg_wall / 1e6:.1f} ms = kernels {kt / 1e6:.1f} ms ({100 * kt / seg_wall:.0f} %) + gaps "
              f"{sum(allg) / 1e6:.1f} ms ({100 * sum(allg) / seg_wall:.0f} %); mean gap {sum(allg) / max(1, len(allg)) / 1e3:.2f} us, median "
              f"{sorted(allg)[len(allg) // 2] / 1e3:.2f} us")
        print("   kernel                                  calls   mean us")
        for k, v in sorted(kern.items(), key=lambda kv: -sum(kv[1])):
            print(f"   {k:40s} {len(v):6d} {sum(v) / len(v) / 1e3:8.2f}")
        print("   gap after -> before                                             n   mean us  median us")
        for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
            v.sort()
            print(f"   {k[0]:30s} -> {k[1]:30s} {len(v):6d} {sum(v) / len(v) / 1e3:8.2f} {v[len(v) // 2] / 1e3:8.2f}")


if __name__ == "__main__":
    main()
