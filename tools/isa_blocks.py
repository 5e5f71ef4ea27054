"""Basic-block instruction census of one kernel in a hipcc -save-temps .s file.
usage: python tools/isa_blocks.py <file.s> <mangled-name-substring> [min_insts]"""
import re, sys, collections
f, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
lines = open(f).read().split('\n')
start = None
for i, l in enumerate(lines):
    if l.startswith(key) and ':' in l:
        start = i; break
assert start is not None, "kernel not found"
blocks = []; cur = [lines[start].split(':')[0], collections.Counter(), 0]
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        break
    m = re.match(r'^(\.LBB[0-9_]+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), collections.Counter(), 0]; continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    cur[1][op] += 1; cur[2] += 1
blocks.append(cur)
tot = sum(b[2] for b in blocks)
valu = sum(n for b in blocks for o, n in b[1].items() if o.startswith('v_'))
print("kernel %s: %d blocks, %d instructions, %d VALU" % (blocks[0][0][:60], len(blocks), tot, valu))
allops = collections.Counter()
for b in blocks: allops.update(b[1])
print("top ops:", ", ".join("%s %d" % kv for kv in allops.most_common(28)))
for b in blocks:
    if b[2] >= minn:
        v = sum(n for o, n in b[1].items() if o.startswith('v_'))
        tr = sum(n for o, n in b[1].items() if o in ('v_rcp_f32', 'v_sqrt_f32', 'v_rsq_f32', 'v_div_scale_f32', 'v_div_fmas_f32', 'v_div_fixup_f32'))
        mem = sum(n for o, n in b[1].items() if o.startswith(('global_', 'scratch_', 'buffer_', 'ds_', 's_load', 's_buffer')))
        print("%-14s insts %4d valu %4d div/sqrt-ops %3d mem %3d  %s" % (b[0], b[2], v, tr, mem, ", ".join("%s %d" % kv for kv in b[1].most_common(6))))
