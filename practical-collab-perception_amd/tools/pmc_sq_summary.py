"""Per-kernel averages of rocprofv3 --pmc counter passes (counter_collection.csv files; one pass per counter group, --kernel-trace only,
program directly after `--`, as MI355X_MICROARCH.md prescribes) with the derived fractions DESIGN.md quotes.
usage: pmc_sq_summary.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]"""
import csv
import json
import sys
from collections import defaultdict

KEYS = [('k_wino4f', 'k_wino4f'), ('k_wino4h', 'k_wino4h'), ('k_wino4c', 'k_wino4c'), ('k_wgrad3x3<', 'k_wgrad3x3'), ('k_conv3x3<2,', 'k_conv3x3_direct<s2>'), ('k_pfn_rows', 'k_pfn_rows'), ('k_pfn<', 'k_pfn'), ('k_sparse_conv_s2', 'k_sparse_conv_s2'), ('k_cell_finish', 'k_cell_finish'), ('k_point_place', 'k_point_place'),
        ('k_conv3x3_wino<1>', 'k_conv3x3_wino<1>'), ('k_w4_gemm', 'k_w4_gemm'), ('k_weight_fuse', 'k_weight_fuse'),
        ('k_stc_scatter', 'k_stc_scatter'), ('k_point_finish', 'k_point_finish'), ('k_point_cells', 'k_point_cells')]


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                for key, label in KEYS:
                    if key in row['Kernel_Name']:
                        acc[label][row['Counter_Name']].append(float(row['Counter_Value']))
                        break
    doc = {'_comment': 'rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline '
                       '--no-overlap (disco, 4 frames x 6 agents x 60k points), one pass per counter group; averages per launch. SQ_WAVE_CYCLES / '
                       'SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles; GRBM_GUI_ACTIVE '
                       'is summed over the 8 XCDs.'}
    for label, counters in acc.items():
        e = {k: round(sum(v) / len(v), 1) for k, v in counters.items()}
        e['launches'] = max(len(v) for v in counters.values())
        d = {}
        wc = e.get('SQ_WAVE_CYCLES')
        if wc:
            for name, key in (('waves_parked_frac (SQ_WAIT_ANY/SQ_WAVE_CYCLES)', 'SQ_WAIT_ANY'),
                              ('issue_stalled_frac (SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)', 'SQ_WAIT_INST_ANY'),
                              ('issuing_frac (SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES)', 'SQ_ACTIVE_INST_ANY')):
                if key in e:
                    d[name] = round(e[key] / wc, 3)
        g = e.get('GRBM_GUI_ACTIVE')
        if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
            d['matrix_pipe_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE/8))'] = round(
                e['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * g / 8), 3)
        if g and 'SQ_LDS_IDX_ACTIVE' in e:
            d['lds_busy_frac_of_kernel (SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE/8))'] = round(e['SQ_LDS_IDX_ACTIVE'] / (256 * g / 8), 3)
        if e.get('SQ_LDS_IDX_ACTIVE') and 'SQ_LDS_BANK_CONFLICT' in e:
            d['lds_bank_conflict_frac_of_lds_cycles'] = round(e['SQ_LDS_BANK_CONFLICT'] / e['SQ_LDS_IDX_ACTIVE'], 3)
        e['derived'] = d
        doc[label] = e
    with open(out_path, 'w') as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: v.get('derived') for k, v in doc.items() if isinstance(v, dict)}, indent=1))


if __name__ == '__main__':
    main()
