"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, separate runs as MI355X_MICROARCH.md prescribes) of
bench.py into per-kernel average fabric-side bytes per launch, keyed the way bench.py's `roofline.kernel` labels start.
usage: pmc_summary.py <config> <fetch counter_collection.csv> <write counter_collection.csv> <out.json>   (out.json is updated in place:
one section per bench config).  FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B); both counters are in KB."""
import csv
import json
import os
import sys
from collections import defaultdict

# kernel label -> the source file whose SHA-256 is recorded next to the counters: bench.py reports `traffic` only while that file is
# unchanged (a kernel edited after the PMC pass would otherwise keep quoting stale bytes)
SOURCES = {'k_wino4f': 'wino4f.hip', 'k_wino4h': 'wino4h.hip', 'k_wino4c': 'wino4c.hip', 'k_conv3x3_wino<1>': 'wino.hip', 'k_conv3x3_wino<2>': 'wino.hip', 'k_wino_ws': 'wino_ws.hip',
           'k_w4_gemm': 'wino4.hip', 'k_w4_input': 'wino4.hip', 'k_w4_output': 'wino4.hip', 'k_conv3x3_direct<s1>': 'conv.hip',
           'k_conv3x3_direct<s2>': 'conv.hip', 'k_pointwise<plain>': 'conv.hip', 'k_pointwise<conv_k2s2>': 'conv.hip',
           'k_pointwise<convT_k2s2>': 'conv.hip', 'k_pfn': 'pfn.hip', 'k_pfn_rows': 'pfn_rows.hip', 'k_point_cells': 'voxelize.hip', 'k_cell_finish': 'voxelize.hip', 'k_point_place': 'voxelize.hip', 'k_stc_scatter': 'voxelize.hip', 'k_sparse_conv_s2': 'sparseconv.hip', 'k_point_head': 'pointhead.hip',
           'k_head_grouped': 'headconv.hip'}
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'csrc')


def source_sha(label):
    import hashlib
    name = SOURCES.get(label)
    path = os.path.join(CSRC, name) if name else None
    if not path or not os.path.isfile(path):
        return None
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


# substring of the rocprof kernel name -> bench.py label key
KEYS = [('k_conv3x3_wino<1>', 'k_conv3x3_wino<1>'), ('k_conv3x3_wino<2>', 'k_conv3x3_wino<2>'), ('k_wino4f', 'k_wino4f'), ('k_wino4h', 'k_wino4h'), ('k_wino4c', 'k_wino4c'), ('k_wino_ws', 'k_wino_ws'),
        ('k_w4_gemm', 'k_w4_gemm'), ('k_w4_input', 'k_w4_input'), ('k_w4_output', 'k_w4_output'),
        ('k_conv3x3<1,', 'k_conv3x3_direct<s1>'), ('k_conv3x3<2,', 'k_conv3x3_direct<s2>'),
        ('k_pointwise<0', 'k_pointwise<plain>'), ('k_pointwise<1', 'k_pointwise<conv_k2s2>'), ('k_pointwise<2', 'k_pointwise<convT_k2s2>'),
        ('k_pfn_rows', 'k_pfn_rows'), ('k_pfn<', 'k_pfn'), ('k_point_cells', 'k_point_cells'), ('k_cell_finish', 'k_cell_finish'), ('k_point_place', 'k_point_place'),
        ('k_stc_scatter', 'k_stc_scatter'), ('k_sparse_conv_s2', 'k_sparse_conv_s2'), ('k_point_head', 'k_point_head'), ('k_head_grouped', 'k_head_grouped')]


def collect(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') != counter:
                continue
            for key, label in KEYS:
                if key in row['Kernel_Name']:
                    acc[label].append(float(row['Counter_Value']))
                    break
    return acc


def main():
    config, out_path = sys.argv[1], sys.argv[4]
    fetch, write = collect(sys.argv[2], 'FETCH_SIZE'), collect(sys.argv[3], 'WRITE_SIZE')
    doc = {}
    if os.path.isfile(out_path):
        with open(out_path) as f:
            doc = json.load(f)
    doc['_comment'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, program directly after `--`) of '
                       '`python3 bench.py --config <section> --steps 3 --warmup 1 --no-cpu-baseline` on MI355X. Units: the counters are KB; '
                       'FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM '
                       'section); Infinity-Cache hits are counted, so this is fabric-side traffic, an upper bound on HBM bytes.')
    sec = {}
    for label in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(label, []), write.get(label, [])
        fa = sum(fv) / max(len(fv), 1)
        wa = sum(wv) / max(len(wv), 1)
        sec[label] = {'launches': len(fv), 'fetch_size_kb_avg': round(fa, 2), 'write_size_kb_avg': round(wa, 2),
                      'bytes_per_launch': round((2.0 * fa + wa) * 1024.0, 2), 'source': SOURCES.get(label), 'source_sha256': source_sha(label)}
    doc[config] = sec
    with open(out_path, 'w') as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(sec, indent=1))


if __name__ == '__main__':
    main()
