"""GPU busy fraction from a rocprofv3 kernel trace: union of the kernel intervals over the span of the trace's last `frac` part
(python gpu_busy.py <kernel_trace.csv> [frac]).  Tells a launch-bound step (gaps between kernels) from a kernel-bound one."""
import csv
import sys


def main(path, frac=0.5):
    iv, names = [], {}
    for r in csv.DictReader(open(path)):
        iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
        names[iv[-1]] = r.get('Kernel_Name', '?')
    iv.sort()
    t_lo = iv[0][0] + (iv[-1][1] - iv[0][0]) * (1.0 - frac)
    iv = [(a, b) for a, b in iv if a >= t_lo]
    span = iv[-1][1] - iv[0][0]
    busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
    gaps, where = [], {}
    last = iv[0]
    for a, b in iv[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            gaps.append(a - cur_b)
            key = (short(names[last]), short(names[(a, b)]))          # the kernel that ended last in front of the gap, the one that ends it
            w = where.setdefault(key, [0, 0])
            w[0] += a - cur_b
            w[1] += 1
            cur_a, cur_b = a, b
            last = (a, b)
        else:
            if b > cur_b:
                last = (a, b)
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    ksum = sum(b - a for a, b in iv)
    gaps.sort()
    print('kernels %d  span %.2f ms  busy (union) %.2f ms = %.3f  kernel-time sum %.2f ms  gaps: n=%d total %.2f ms median %.1f us p90 %.1f us'
          % (len(iv), span / 1e6, busy / 1e6, busy / span, ksum / 1e6, len(gaps), sum(gaps) / 1e6,
             gaps[len(gaps) // 2] / 1e3 if gaps else 0, gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0))
    for (k0, k1), (t, n) in sorted(where.items(), key=lambda kv: -kv[1][0])[:12]:
        print('   idle %7.2f ms in %4d gaps  after %-40s before %s' % (t / 1e6, n, k0, k1))


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:40]


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
