#!/bin/bash
# A/B: workgroup-level dynamic row hand-out (edge_row_dealing = 2) in the key pass
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03l
python - <<'PY'
import torch, bench, argparse
from targetdiff_amd import workloads
dev = torch.device('cuda:0')
outs = []
for opt in (1, 2):
    bargs = argparse.Namespace(knn=32, cutoff_mode='knn', radius=6.0, cap=32, fp32_node_gemms=False, option=[f'edge_row_dealing={opt}'])
    model = bench.build_model(dev, bargs)
    pockets, spp, sizes, desc = bench.make_workload('c2', 0)
    batch = workloads.pack_samples(pockets, spp, sizes).to(dev)
    gen = torch.Generator(device='cpu').manual_seed(2021)
    lpos, lv = workloads.init_ligand(workloads.pack_samples(pockets, spp, sizes), generator=gen, spread=2.0)
    out = model(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos.to(dev), lv.to(dev), batch.ligand_element_batch)
    outs.append({k: v.clone() for k, v in out.items()})
for k in outs[0]:
    print(k, 'bit-identical' if torch.equal(outs[0][k], outs[1][k]) else 'DIFFERENT', float((outs[0][k] - outs[1][k]).abs().max()))
PY
for O in 1 2 1 2; do python bench.py --no-cpu-baseline --no-full-run --no-stateless --profile-all --option edge_row_dealing=$O > gpurun_out/r03l/c2_deal$O.json 2> gpurun_out/r03l/c2_deal${O}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/r03l/c2_deal$O.json')); print('deal=$O', round(d['ms_per_step'],3))"; grep "x2h_k\|x2h_v" gpurun_out/r03l/c2_deal${O}_breakdown.txt; done
python tools/wg_balance.py --option edge_row_dealing=2 2>&1 | grep -v "^/opt" > gpurun_out/r03l/wg_balance_deal2.txt; head -12 gpurun_out/r03l/wg_balance_deal2.txt
grep "value" gpurun_out/r03l/wg_balance_deal2.txt
for W in c1 c3 c5; do for O in 1 2; do python bench.py --workload $W --no-cpu-baseline --no-stateless --option edge_row_dealing=$O > gpurun_out/r03l/${W}_deal$O.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03l/${W}_deal$O.json')); print('$W deal=$O', round(d['ms_per_step'],3))"; done; done
for O in 1 2; do python bench.py --workload c5 --no-cpu-baseline --no-stateless --knn 48 --option edge_row_dealing=$O > gpurun_out/r03l/c5k48_deal$O.json 2>/dev/null; python bench.py --workload c5 --no-cpu-baseline --no-stateless --cutoff-mode hybrid --option edge_row_dealing=$O > gpurun_out/r03l/c5hyb_deal$O.json 2>/dev/null; python -c "
import json
for n in ('c5k48','c5hyb'):
    d=json.load(open('gpurun_out/r03l/%s_deal$O.json' % n)); print(n, 'deal=$O', round(d['ms_per_step'],3))"; done
