"""profiles/r3_olbm_lane.txt, tables part, from gpurun_out/lane_grid.txt (tools/olbm_lane_grid.sh), gpurun_out/lane_wide.txt /
lane_wide_before.txt (tools/olbm_cols_probe.py at p = 65 .. 256 with the lane route on / off) and the counter summaries of
tools/olbm_lane_pmc.sh under gpurun_out/lanepmc<p>/.  usage: python tools/olbm_lane_report.py > tables.txt"""
import json
import re


def parse(fn):
    rows = []
    for ln in open(fn):
        m = re.search(r'len=(\d+) p=\s*(\d+)\s+(\w+)\s+kernel\s+([\d.]+) ms\s+call\s+([\d.]+) ms\s+input\s+([\d.]+)', ln)
        if m:
            rows.append((int(m.group(1)), int(m.group(2)), m.group(3), float(m.group(4)), float(m.group(5)), float(m.group(6))))
    return rows


sec, grid = None, {}
for ln in open('gpurun_out/lane_grid.txt'):
    if ln.startswith('=='):
        m = re.match(r'== n len = (\d+) (\d+), (\w+)', ln)
        sec = (int(m.group(1)), int(m.group(2)), m.group(3))
        grid[sec] = []
    else:
        m = re.search(r'p=\s*(\d+)\s+(\w+)\s+kernel\s+([\d.]+) ms\s+call\s+([\d.]+) ms\s+input\s+([\d.]+)', ln)
        if m:
            grid[sec].append((int(m.group(1)), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5))))
HEAD = "   p   kernel            ms   (call)   input TB/s |  before: kernel            ms   (call)\n"


def table(n, ln):
    a, b = grid[(n, ln, 'lane')], {r[0]: r for r in grid[(n, ln, 'before')]}
    s = HEAD
    for r in a:
        q = b.get(r[0])
        s += " %3d   %-17s %7.3f (%7.3f)   %5.2f     |  %-17s %8.3f (%7.3f)\n" % (r[0], r[1], r[2], r[3], r[4], q[1] if q else '', q[2] if q else 0, q[3] if q else 0)
    return s


out = ["== n = 1e8, len = 1e4: kernel ms (call ms)\n", table(100000000, 10000)]
wide, wbd = parse('gpurun_out/lane_wide.txt'), {r[1]: r for r in parse('gpurun_out/lane_wide_before.txt')}
s = HEAD
for r in wide:
    q = wbd.get(r[1])
    s += " %3d   %-17s %7.3f (%7.3f)   %5.2f     |  %s\n" % (r[1], r[2], r[3], r[4], r[5], ("%-17s %8.3f (%7.3f)" % (q[2], q[3], q[4])) if q else '')
out.append(s)
for (n, ln) in ((100000000, 1000), (100000000, 100000), (100000000, 1000000), (10000000, 1000), (1000000, 1000), (200000, 512)):
    out.append("\n== n = %d, len = %d\n" % (n, ln))
    out.append(table(n, ln))
out.append("\n== counters (tools/olbm_lane_pmc.sh: rocprofv3 --kernel-trace --pmc, a pass per group; n = 1e8, len = 1e4; FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes)\n")
for p in (16, 64, 128, 192):
    u, f, i, w = (json.load(open('gpurun_out/lanepmc%d/%s.json' % (p, k))) for k in ('util', 'fetch', 'inst', 'write'))
    c, ms = u['per_launch_counters'], u['avg_launch_ms']
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    fb = f['per_launch_counters']['FETCH_SIZE'] * 2048 / 1e9
    out.append("p=%3d  %.3f ms at %.2f GHz: matrix pipe busy %.1f %% of the cycles; a wave waits %.0f %% of its cycles; HBM fetch %.2f GB = %.3f x the input at %.2f TB/s; written %.3f GB; %.1f vector instructions per MFMA (itself included); LDS bank conflicts %d\n" % (
        p, ms, cyc / ms / 1e6, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc, 100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], fb,
        fb / (8 * 99999744 * p / 1e9), fb / f['avg_launch_ms'], w['per_launch_counters']['WRITE_SIZE'] * 2048 / 1e9,
        c['SQ_INSTS_VALU'] / (c['SQ_INSTS_VALU_MFMA_MOPS_F64'] / 4), i['per_launch_counters']['SQ_LDS_BANK_CONFLICT']))
print(''.join(out), end='')
