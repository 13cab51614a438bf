#!/usr/bin/env python3
"""End to end on synthetic data, one GPU: a packed split resident in HBM -> RetrieverTrainer.fit (the reference's training
defaults: dropout 0.1, hide-and-seek, AdamW 1e-3 / 1e-4, cosine schedule, gradient clipping at 1.0) -> a Lightning-layout
checkpoint -> strict load into a fresh Retriever (what src/eval.py does) -> RetrieverEvaluator with the top-k artifact writer.

    python examples/train_and_eval_synthetic.py [--graphs 256] [--dim 64] [--epochs 5] [--out /tmp/evi_example] [--precision bf16-mixed]

Everything between the loader and the metric dictionary runs in hand-written HIP kernels (libevi_hip.so); there is no CPU path.
Under `python -m torch.distributed.run --nproc-per-node N` each rank trains on its share of the graphs and the flat gradient
is averaged with one RCCL all-reduce per step.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evi_rag_amd import eval as ev  # noqa: E402
from evi_rag_amd import packed_dataset as pd, synthetic  # noqa: E402
from evi_rag_amd.embedding_store import GlobalEmbeddingStore  # noqa: E402
from evi_rag_amd.eval_loop import RetrieverEvaluator  # noqa: E402
from evi_rag_amd.loss import RetrieverLoss  # noqa: E402
from evi_rag_amd.retriever import Retriever  # noqa: E402
from evi_rag_amd.train import RetrieverTrainer  # noqa: E402

HIDE_SEEK = {"enabled": True, "p_near": 0.7, "p_far": 0.1, "bias_near": -2.0, "bias_far": -0.5, "apply_in_eval": False}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=256)
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--edges", type=int, default=600)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--relations", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--out", default="/tmp/evi_example")
    ap.add_argument("--precision", default="32-true", help="trainer.precision: 32-true (split-bf16, f32-grade GEMMs) or bf16-mixed (one bf16 product)")
    args = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    os.makedirs(args.out, exist_ok=True)

    # data: every rank writes / reads the same synthetic split (seeded); PackedLoader hands each rank its graphs
    D = args.dim
    base = synthetic.make_batch(args.graphs, nodes_per_graph=args.nodes, edges_per_graph=args.edges, emb_dim=D,
                                num_relations=args.relations, seed=1, attach_embeddings=False)
    split = os.path.join(args.out, f"train_rank{rank}.packed")
    pd.write_packed(split, pd.samples_from_flat_batch(base))
    rng = np.random.default_rng(0)
    store = GlobalEmbeddingStore.from_tensors(torch.from_numpy(rng.standard_normal((int(base.node_embedding_ids.max()) + 1, D)).astype(np.float32)).to(dev),
                                              torch.from_numpy(rng.standard_normal((args.relations, D)).astype(np.float32)).to(dev), device=dev)
    ds = pd.PackedRetrievalDataset(split, device=dev, embeddings=store)

    torch.manual_seed(0)
    model = Retriever(emb_dim=D, hidden_dim=D, dropout_p=0.1, hide_seek_cfg=HIDE_SEEK).to(dev)
    loss = RetrieverLoss(infonce_weight=1.0, bce_weight=0.0)  # configs/experiment/train_retriever.yaml
    trainer = RetrieverTrainer(model, loss=loss, optimizer_cfg={"type": "adamw", "lr": 1e-3, "weight_decay": 1e-4},
                               scheduler_cfg={"type": "cosine", "t_max": max(args.epochs, 1), "eta_min": 1e-6}, gradient_clip_val=1.0,
                               precision=args.precision)
    log = trainer.fit(pd.PackedLoader(ds, batch_size=args.batch_size, shuffle=True, random_seed=0, rank=rank, world_size=world),
                      max_epochs=args.epochs)
    if rank == 0:
        for i, e in enumerate(log["epochs"]):
            print(f"epoch {i}: train/loss {e['train/loss']:.4f}  lr {e['lr']:.2e}")
        print(f"{log['steps']} steps in {log['seconds']:.2f} s ({log['steps'] * args.batch_size / log['seconds']:.0f} questions/s per rank)")
        ckpt = os.path.join(args.out, "last.ckpt")
        trainer.save_checkpoint(ckpt)

        # evaluation from the checkpoint, like src/eval.py: strict load into a fresh module, metrics + top-k artifact
        fresh = Retriever(emb_dim=D, hidden_dim=D, dropout_p=0.1, hide_seek_cfg=HIDE_SEEK).to(dev).eval()
        ev.load_checkpoint_strict(fresh, ckpt)
        res = RetrieverEvaluator(fresh, loss=loss, k_values=[1, 10, 50, 100]).run(pd.PackedLoader(ds, batch_size=args.batch_size))
        keep = {k: round(v, 4) for k, v in res["metrics"].items() if k.endswith(("@10", "@100", "loss"))}
        print(json.dumps({"questions_per_s": round(res["queries_per_sec"]), **keep}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
