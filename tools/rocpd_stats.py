"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table.

usage: python tools/rocpd_stats.py <results.db> [--skip-first N] > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('void ', '').replace('d4::', '')
    return name[:70]


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    sfx = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
    rows = list(cur.execute(
        f"select s.kernel_name, d.end - d.start, d.start, d.end, d.grid_size_x, d.workgroup_size_x, s.arch_vgpr_count, s.accum_vgpr_count, d.group_segment_size "
        f"from rocpd_kernel_dispatch{sfx} d join rocpd_info_kernel_symbol{sfx} s on d.kernel_id = s.id order by d.start"))
    agg = {}
    for name, dur, st, en, gx, wx, vg, ag, lds in rows:
        a = agg.setdefault(short(name), dict(n=0, t=0, mn=1 << 62, mx=0, vg=vg, ag=ag, lds=lds))
        a['n'] += 1; a['t'] += dur; a['mn'] = min(a['mn'], dur); a['mx'] = max(a['mx'], dur)
    total = sum(a['t'] for a in agg.values())
    print(f'# {path}: {len(rows)} dispatches, total kernel time {total / 1e6:.3f} ms')
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['t']):
        print(f"{k:72s} {a['n']:7d} {a['t'] / 1e6:10.3f} {a['t'] / a['n'] / 1e3:9.2f} {a['mn'] / 1e3:9.2f} {a['mx'] / 1e3:9.2f} {100 * a['t'] / total:6.2f} {a['vg']:5d} {a['ag']:5d} {a['lds']:7d}")
    # idle time between consecutive dispatches (start of i+1 minus end of i), attributed to the kernel that FOLLOWS the gap
    gaps = {}
    tot_gap = 0
    span = rows[-1][3] - rows[0][2]
    for a, b in zip(rows[:-1], rows[1:]):
        g = b[2] - a[3]
        if g > 200000:          # host-side pause (> 0.2 ms): not a launch gap
            span -= g
            continue
        tot_gap += g
        d = gaps.setdefault(short(b[0]), [0, 0])
        d[0] += 1; d[1] += g
    print(f'# inter-kernel idle: {tot_gap / 1e6:.3f} ms over {len(rows) - 1} gaps = {tot_gap / max(len(rows) - 1, 1) / 1e3:.2f} us avg; busy span {span / 1e6:.3f} ms')
    for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f'#   before {k:60s} {n:6d} gaps  avg {t / n / 1e3:7.2f} us  total {t / 1e6:8.3f} ms')


if __name__ == '__main__':
    main()
