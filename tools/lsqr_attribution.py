#!/usr/bin/env python3
"""Where the wall time of ONE spring-inpaint solve goes, from a rocprofv3 kernel trace with timestamps.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_lsqr -- python3 $R/tools/pmc_lsqr_run.py
    python tools/lsqr_attribution.py $OUT/trace_lsqr/**/*_kernel_trace.csv [out.md]

The solve is the span from its first kernel (mask_kernel / setup_kernel) to its last (scatter_kernel).  Every nanosecond of
that span is either inside a kernel or in the gap before the next one; gaps are attributed to the kernel they precede:
  set-up        the plane passes before the first iteration (mask, rhs, atu, w_init, the first av) and their scalar launches
  streaming     atu_kernel + xwav_kernel of real iterations (the two plane passes LSQR needs per iteration)
  scalar        reduce_scalar_kernel launches of real iterations (one block: the partial sums + the scalar recurrence)
  past the stop launches enqueued after the stopping iteration, between two host polls (they return at once)
  tails         gap before a kernel that follows another kernel of the same chunk (dispatch of a dependent launch)
  polls         gap before the first kernel after a host poll (hipMemcpyAsync + hipStreamSynchronize + re-enqueue)
"""
import csv
import sys


def short(name):
    for k in ("atuxw_kernel", "av2_kernel", "s_count_bnorm", "mask_kernel", "rhs_kernel", "setup_kernel", "atu_kernel", "xwav_kernel", "av_kernel", "w_init_kernel", "scatter_kernel",
              "reduce_scalar_kernel", "reduce2_kernel", "reduce_kernel", "s_count", "s_bnorm", "s_init_alfa", "xw_kernel",
              "__amd_rocclr_copyBuffer"):
        if k in name:
            return k
    return None


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Kernel_Name"]))
    rows.sort()
    first = next(i for i, r in enumerate(rows) if r[2] in ("mask_kernel", "setup_kernel"))
    last = max(i for i, r in enumerate(rows) if r[2] == "scatter_kernel")
    ks = rows[first:last + 1]
    assert all(k[2] for k in ks), [k[3][:60] for k in ks if not k[2]]
    span = ks[-1][1] - ks[0][0]
    # iterations: an atu_kernel after the set-up's own (the first one) starts iteration i; durations tell real from stopped
    split = any(k[2] == "atuxw_kernel" for k in ks)      # round 5's iteration: atuxw | reduce | av2 | reduce
    if split:
        atu = [next(i for i, k in enumerate(ks) if k[2] == "atu_kernel")] + [i for i, k in enumerate(ks) if k[2] == "atuxw_kernel"]
    else:
        atu = [i for i, k in enumerate(ks) if k[2] == "atu_kernel"]
    durs = sorted(ks[i][1] - ks[i][0] for i in atu[1:])
    typical = durs[len(durs) // 2]
    real_iters = [i for i in atu[1:] if ks[i][1] - ks[i][0] > 0.2 * typical]
    n_real = len(real_iters)
    end_real = real_iters[-1] + 1                          # index after the last real iteration's 4 launches
    seen = 0
    while seen < 3:
        seen += ks[end_real][2] != "__amd_rocclr_copyBuffer"
        end_real += 1
    cat = {}

    def add(name, ns):
        cat[name] = cat.get(name, 0) + ns

    for i, k in enumerate(ks):
        d = k[1] - k[0]
        gap = k[0] - ks[i - 1][1] if i else 0
        if i < atu[1]:
            add("set-up kernels", d)
            add("set-up gaps (incl. the 2 host syncs before the loop)", gap)
        elif k[2] == "scatter_kernel":
            add("scatter", d)
            add("gap before scatter (last poll)", gap)
        elif i >= end_real:
            add("launches past the stop", d + gap)
        else:
            if k[2] in ("atu_kernel", "xwav_kernel", "atuxw_kernel", "av2_kernel"):
                add("streaming kernels (atu + xwav | atuxw + av2), real iterations", d)
            elif k[2] == "__amd_rocclr_copyBuffer":
                add("polls (gap > 20 us before a kernel: host readback + re-enqueue)", d)
            else:
                add("scalar kernels (reduce + recurrence), real iterations", d)
            if gap > 20000:
                add("polls (gap > 20 us before a kernel: host readback + re-enqueue)", gap)
                cat["n_polls"] = cat.get("n_polls", 0) + 1
            else:
                add("tails (gap before a dependent launch)", gap)
    n_polls = cat.pop("n_polls", 0)
    lines = ["| where | ms | share | note |", "|---|---|---|---|"]
    tot = 0
    for name, ns in sorted(cat.items(), key=lambda kv: -kv[1]):
        tot += ns
        lines.append("| %s | %.3f | %.1f %% | |" % (name, ns / 1e6, 100.0 * ns / span))
    lines.append("| **solve span (first kernel start to scatter end)** | **%.3f** | 100 %% | %d real iterations, %d launched; %d polls inside the loop |"
                 % (span / 1e6, n_real, len(atu) - 1, n_polls))
    per = {}
    for k in ks[atu[1]:end_real]:
        per.setdefault(k[2], []).append(k[1] - k[0])
    lines.append("")
    lines.append("Per real iteration: " + ", ".join("%s %.1f us" % (n, sum(v) / len(v) / 1e3) for n, v in sorted(per.items()))
                 + "; span / iteration %.3f ms" % (span / 1e6 / max(1, n_real)))
    pre = {}
    for k in ks[:atu[1]]:
        pre.setdefault(k[2], []).append(k[1] - k[0])
    lines.append("Set-up kernels: " + ", ".join("%s %.1f us" % (n, sum(v) / 1e3) for n, v in pre.items()))
    assert abs(tot - span) < 1000, (tot, span)
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
