"""Merge rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv per pass) into a per-kernel table.

usage: python tools/pmc_summary.py <dir_pass1> [<dir_pass2> ...] [--json profiles/pmc_traffic.json] > profiles/<name>.txt

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced read stream -> it is doubled here
("fetch_x2"); WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('void ', '').replace('d4::', '')
    return name[:52]


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-dispatch values
    dur = collections.defaultdict(list)
    args = list(sys.argv[1:])
    json_path = None
    if '--json' in args:
        i = args.index('--json')
        json_path = args[i + 1]
        del args[i:i + 2]
    for d in args:
        cc = glob.glob(os.path.join(d, '*counter_collection.csv'))[0]
        kt = glob.glob(os.path.join(d, '*kernel_trace.csv'))[0]
        trace = {r['Dispatch_Id']: r for r in csv.DictReader(open(kt))}
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(cc)):
            acc[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
        for (did, name), cs in acc.items():
            k = short(name)
            for c, v in cs.items():
                per[k][c].append(v)
            t = trace.get(did)
            if t:
                dur[(k, d)].append(int(t['End_Timestamp']) - int(t['Start_Timestamp']))
    counters = sorted({c for k in per for c in per[k]})
    print('# per-dispatch averages; duration = mean over the passes (profiled clocks are ~3 % lower than unprofiled)')
    hdr = f"{'kernel':54s} {'n':>6s} {'dur_us':>9s}"
    for c in counters:
        hdr += f' {c[:22]:>22s}'
    hdr += f" {'HBM_GB/s(fetch_x2+write)':>26s} {'MFMA_busy_%':>12s}"
    print(hdr)
    rows = []
    for k in per:
        ds = [x for (kk, d), v in dur.items() if kk == k for x in v]
        mean_dur = sum(ds) / max(len(ds), 1)
        n = max(len(v) for v in per[k].values())
        avg = {c: (sum(per[k][c]) / len(per[k][c]) if per[k][c] else float('nan')) for c in counters}
        gbs = float('nan')
        if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg and mean_dur > 0:
            gbs = (2 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024 / mean_dur      # bytes / ns = GB/s
        mfma = float('nan')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and 'GRBM_GUI_ACTIVE' in avg and avg['GRBM_GUI_ACTIVE'] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs each with one matrix pipe
            mfma = 100. * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] / 8 * 1024)
        rows.append((mean_dur * n, k, n, mean_dur, avg, gbs, mfma))
    for _, k, n, md, avg, gbs, mfma in sorted(rows, reverse=True):
        line = f'{k:54s} {n:6d} {md / 1e3:9.2f}'
        for c in counters:
            line += f' {avg[c]:22.1f}'
        line += f' {gbs:26.1f} {mfma:12.1f}'
        print(line)
    if json_path:
        import json
        out = dict(method='rocprofv3 --pmc FETCH_SIZE (+GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES) and --pmc WRITE_SIZE (+SQ_WAIT_INST_ANY '
                          'SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY) in separate passes over tools/rollout_profile_target.py (4 frames of the cfg-2 rollout, tile '
                          'choices preloaded from the tuning cache so no timing launches are counted); KiB units; FETCH_SIZE doubled (gfx950 reports 1/2 of '
                          'a wide coalesced read stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported', kernels={})
        for _, k, n, md, avg, gbs, mfma in sorted(rows, reverse=True):
            if md * n < 50e3 or 'FETCH_SIZE' not in avg or 'WRITE_SIZE' not in avg or k.startswith(('at::', '__amd')):
                continue
            out['kernels'][k] = dict(launches=n, avg_us=round(md / 1e3, 2), fetch_size_kib_per_launch=round(avg['FETCH_SIZE'], 1),
                                     write_size_kib_per_launch=round(avg['WRITE_SIZE'], 1),
                                     hbm_bytes_per_launch=int((2 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024), hbm_gbs=round(gbs, 1),
                                     mfma_busy_pct=None if mfma != mfma else round(mfma, 1))
        json.dump(out, open(json_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
