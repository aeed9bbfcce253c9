"""Fold the PMC summaries written by tools/pmc_sets.sh (… full) into profiles/pmc_traffic.json entries.
usage: python tools/pmc_json.py <key> <summary.txt> <kernel-name substring> "<workload note>" <algorithmic_bytes> [expected_mfma]"""
import json, os, re, sys
key, path, sub, note, alg = sys.argv[1:6]
exp_mfma = float(sys.argv[6]) if len(sys.argv) > 6 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cur, vals = None, {}
for line in open(path):
    if line.startswith('== '):
        cur = line[3:].strip()
    m = re.match(r'\s+(\S+)\s+n=\d+ avg ([\d.e+-]+)', line)
    if m and cur and sub in cur:
        vals[m.group(1)] = float(m.group(2))
assert vals, 'kernel %r not found in %s' % (sub, path)
rec = {'workload': note, 'source': os.path.relpath(path, root), 'algorithmic_bytes': int(float(alg))}
# which tree the counters describe (bench.py prints it beside the live timing so a stale record is visible): the GPU box has no
# .git, so the commit travels in .fami_sha (written by tools/gpu.sh before the snapshot) or FAMI_GIT_SHA
sha = os.environ.get('FAMI_GIT_SHA')
if not sha and os.path.exists(os.path.join(root, '.fami_sha')):
    sha = open(os.path.join(root, '.fami_sha')).read().strip()
rec['git_sha'] = sha
if 'FETCH_SIZE' in vals:
    rec['fetch_kb'] = vals['FETCH_SIZE']; rec['read_bytes_corrected'] = int(vals['FETCH_SIZE'] * 1024 * 2)
if 'WRITE_SIZE' in vals:
    rec['write_kb'] = vals['WRITE_SIZE']; rec['write_bytes'] = int(vals['WRITE_SIZE'] * 1024)
if 'SQ_INSTS_MFMA' in vals:
    rec['sq_insts_mfma'] = vals['SQ_INSTS_MFMA']
    if exp_mfma:
        rec['expected_mfma_insts'] = exp_mfma
if 'SQ_VALU_MFMA_BUSY_CYCLES' in vals and 'GRBM_GUI_ACTIVE' in vals:
    cyc = vals['GRBM_GUI_ACTIVE'] / 8.0                      # summed over the 8 XCDs
    rec['kernel_cycles'] = cyc
    rec['sq_valu_mfma_busy_cycles'] = vals['SQ_VALU_MFMA_BUSY_CYCLES']
    rec['mfma_busy'] = round(vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc, 4)      # 1024 SIMDs
for k in ('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'TCC_HIT_sum', 'TCC_MISS_sum'):
    if k in vals:
        rec[k.lower()] = vals[k]
jp = os.path.join(root, 'profiles', 'pmc_traffic.json')
d = json.load(open(jp))
d[key] = rec
json.dump(d, open(jp, 'w'), indent=1)
print(key, json.dumps(rec))
