"""Static instruction mix of the k_tile variants in a hipcc -S listing: python scripts/isa_mix.py k.s [waves]"""
import re, collections, sys
txt = open(sys.argv[1]).read().splitlines()
waves = sys.argv[2] if len(sys.argv) > 2 else "4"
cur = None; ops = {}
for line in txt:
    m = re.match(r'^(_ZN2tr12_GLOBAL__N_16k_tileILi(\d)ELi(\d+)EEEvNS_8TileArgsE):', line)
    if m:
        cur = (int(m.group(2)), m.group(3)); ops[cur] = collections.Counter(); continue
    if cur and line.strip().startswith('s_endpgm'):
        pass
    if cur and line.startswith('.Lfunc_end'):
        cur = None; continue
    if cur:
        mm = re.match(r'\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|scratch_[a-z0-9_]+)', line)
        if mm: ops[cur][mm.group(1)] += 1
for (fs, w), c in sorted(ops.items()):
    if w != waves: continue
    cls = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_pk_'): cls['v_pk'] += v
        elif '_f64' in k: cls['f64'] += v
        elif re.match(r'v_(div_|rcp|rsq|sqrt)', k): cls['div/rcp/sqrt'] += v
        elif k.startswith('v_readlane') or k.startswith('v_readfirstlane'): cls['readlane'] += v
        elif k.startswith('v_'): cls['v_other'] += v
        elif k.startswith('s_'): cls['salu'] += v
        elif k.startswith('scratch'): cls['scratch'] += v
        else: cls['mem'] += v
    print("FS %d waves %s: total %5d %s" % (fs, w, sum(c.values()), dict(cls)))
    if len(sys.argv) > 3 and int(sys.argv[3]) == fs:
        for k, v in c.most_common(40): print("     %-28s %d" % (k, v))
