"""profiles/rNN_exchange_n2_one_gpu.json from the two bench lines of the N = 2 rehearsal on one GPU:
    python scripts/summarise_n2.py rNN gpurun_out/n2_peer.json gpurun_out/n2_sparse.json"""
import json, sys
tag, dense, sparse = sys.argv[1:4]
KEEP = ["metric", "value", "ms_per_step", "n_gpus", "steps", "config", "per_rank", "exchange_bytes_per_frame",
        "exchange_dense_bytes_per_frame", "parity_vs_oracle", "scale_config"]
def line(path):
    for l in open(path):
        if l.startswith("{"):
            d = json.loads(l)
            return {k: d.get(k) for k in KEEP}
    raise SystemExit("no bench line in " + path)
out = {"dense (bands pulled by the DMA engines)": line(dense), "sparse (k_push_tiles)": line(sparse),
       "what": "bench.py --gpus 2 --exchange peer [--sparse] --steps 40 --warmup 8 with BOTH rank processes on one MI355X "
               "(TR_BENCH_SHARE_GPU=1): a functional rehearsal of the N > 1 path with a real peer (band scenes, groups, exchange, "
               "assembled frame = oracle), NOT a speed claim: the ranks share the GPU's compute units and exchange through its "
               "memory, not xGMI"}
json.dump(out, open("profiles/%s_exchange_n2_one_gpu.json" % tag, "w"), indent=1)
print("wrote profiles/%s_exchange_n2_one_gpu.json" % tag)
