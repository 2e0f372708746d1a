"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, separate runs as MI355X_MICROARCH.md prescribes) of
bench.py into per-kernel average fabric-side bytes per launch.  usage: pmc_summary.py <fetch counter_collection.csv> <write ...csv> <out.json>
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B); both counters are in KB."""
import csv
import json
import sys
from collections import defaultdict

KEYS = {'k_w4_gemm': 'k_w4_gemm', 'k_w4_input': 'k_w4_input', 'k_w4_output': 'k_w4_output', 'k_conv3x3_wino<2>': 'k_conv3x3_wino<2>',
        'k_conv3x3_wino<1>': 'k_conv3x3_wino<1>'}


def collect(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') != counter:
                continue
            for key, label in KEYS.items():
                if key in row['Kernel_Name']:
                    acc[label].append(float(row['Counter_Value']))
    return acc


def main():
    fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
    out = {'_comment': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --steps 3 --warmup 1 '
                       '--no-cpu-baseline --conv-algo auto` on MI355X. Units: the counters are KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests '
                       'at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted, so this is '
                       'fabric-side traffic, an upper bound on HBM bytes.'}
    for label in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(label, []), write.get(label, [])
        fa = sum(fv) / max(len(fv), 1)
        wa = sum(wv) / max(len(wv), 1)
        out[label] = {'launches': len(fv), 'fetch_size_kb_avg': round(fa, 2), 'write_size_kb_avg': round(wa, 2),
                      'bytes_per_launch': round((2.0 * fa + wa) * 1024.0, 2)}
    with open(sys.argv[3], 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
