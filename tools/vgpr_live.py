"""Approximate VGPR liveness along one kernel of a gfx950 .s listing (straight-line approximation:
loops and branches are ignored).  usage: vgpr_live.py file.s kernel-substring [window]"""
import re, sys

def regs_in(t):
    r = set(int(x) for x in re.findall(r'\bv(\d+)\b', t))
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
        r |= set(range(int(a), int(b) + 1))
    return r

def main():
    path, sub = sys.argv[1], sys.argv[2]
    win = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines, on = [], False
    for l in open(path):
        if not on and re.match(r'^_Z\w+:', l) and sub in l: on = True
        if on:
            lines.append(l.rstrip('\n'))
            if 's_endpgm' in l: break
    cur, ranges = {}, []
    for i, l in enumerate(lines):
        code = l.split(';')[0]
        m = re.match(r'\s+([a-z_0-9]+)\s+(.*)', code)
        if not m: continue
        op, args = m.groups()
        parts = args.split(',')
        nodst = op.startswith(('ds_write', 'global_store', 's_', 'buffer_store', 'v_cmp', 'v_cmpx', 'scratch_store'))
        dst = set() if nodst else regs_in(parts[0])
        src = regs_in(args) if nodst else regs_in(','.join(parts[1:]))
        for r in src:
            if r in cur: cur[r][1] = i
        for r in dst:
            if r in cur: ranges.append(tuple(cur[r]))
            cur[r] = [i, i]
    ranges += [tuple(v) for v in cur.values()]
    n = len(lines)
    d = [0] * (n + 2)
    for a, b in ranges:
        d[a] += 1; d[b + 1] -= 1
    acc, live = 0, []
    for i in range(n):
        acc += d[i]; live.append(acc)
    for i in range(0, n, win):
        seg = lines[i:i + win]
        k = {}
        for l in seg:
            m = re.match(r'\s+([a-z_0-9]+)', l)
            if m:
                op = m.group(1)
                t = ('ds_r' if op.startswith('ds_read') else 'ds_w' if op.startswith('ds_write') else
                     'gld' if op.startswith('global_load') else 'gst' if op.startswith('global_store') else
                     'sld' if op.startswith('s_load') else 'valu' if op.startswith('v_') else None)
                if t: k[t] = k.get(t, 0) + 1
        print("%5d  live<=%3d  %s" % (i, max(live[i:i + win]), k))

main()
