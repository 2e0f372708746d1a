"""GPU busy / idle share of the steady-state steps from a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end] intervals against
the wall window they span.  usage: gpu_idle.py <kernel_trace.csv> [window start, ms before the last kernel = 250] [window end = 60]
(bench.py --steps 20: the window lies inside the timed loop, after model build / warm-up and before the instrumented pass)"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    a = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
    b = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
    t1 = max(e for _s, e, _n in iv)
    iv = [x for x in iv if t1 - a * 1e6 <= x[0] <= t1 - b * 1e6]
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e, n in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = cur_e - iv[0][0]
    print('window %.2f ms, GPU busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps' % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps)))
    big = sorted(gaps, reverse=True)[:25]
    print('largest gaps (us, kernel that ended the gap):')
    for g, n in big:
        print('  %8.1f  %s' % (g / 1e3, n[:100]))
    import collections
    by = collections.Counter()
    for g, n in gaps:
        by[n[:70]] += g
    print('idle time by the kernel that follows the gap:')
    for n, g in by.most_common(15):
        print('  %8.2f ms  %s' % (g / 1e6, n))


if __name__ == '__main__':
    main()
