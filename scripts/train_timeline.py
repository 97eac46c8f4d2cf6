"""Timeline of ONE training step from a rocprofv3 kernel trace (scripts/train_trace.sh): every kernel of the last complete step in start
order with its queue, start offset and duration, the gaps on the main queue and the busy time of each queue.  Shows what the SUMMED table
of summarize_train_trace.py cannot: which kernels run beside which, and where the critical path waits.
usage: python scripts/train_timeline.py gpurun_out/<tag>/train/t_kernel_trace.csv [--all]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("fn2::", "")
    n = re.sub(r"at::native::.*multi_tensor_apply_kernel.*", "torch fused optimizer", n)
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "at::", n)
    return n[:78]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    rows.sort()
    # a step ends with the fused optimizer's last launch; take the last two optimizer groups
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[3]]
    ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] != i + 1 and rows[opt[k + 1]][0] - rows[i][1] > 1_000_000]
    if len(ends) < 2:
        print("no two optimizer steps in the trace")
        return
    a, b = ends[-2] + 1, ends[-1] + 1
    step = rows[a:b]
    t0 = step[0][0]
    print("step: %d kernels, %.3f ms from first start to last end" % (len(step), (step[-1][1] - t0) / 1e6))
    queues = sorted({r[2] for r in step})
    main_q = max(queues, key=lambda q: sum(1 for r in step if r[2] == q))
    for q in queues:
        ks = [r for r in step if r[2] == q]
        busy = sum(r[1] - r[0] for r in ks)
        print("  queue %d%s: %d kernels, busy %.3f ms" % (q, " (main)" if q == main_q else "", len(ks), busy / 1e6))
    last_end = {q: None for q in queues}
    gaps = 0
    show_all = "--all" in sys.argv
    for s, e, q, n in step:
        gap = (s - last_end[q]) / 1e3 if last_end[q] is not None else 0.0
        if q == main_q and gap > 0:
            gaps += gap
        others = [r for r in step if r[2] != q and r[0] < e and r[1] > s]
        if show_all or (e - s) > 40_000 or gap > 15:
            print("%9.1f us  q%d  %8.1f us  gap %6.1f  %-78s %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, gap, short(n),
                                                                      ("| beside: " + ", ".join(short(o[3])[:28] for o in others[:2])) if others else ""))
        last_end[q] = max(e, last_end[q] or 0)
    print("idle gaps on the main queue: %.3f ms" % (gaps / 1e3))


if __name__ == "__main__":
    main()
