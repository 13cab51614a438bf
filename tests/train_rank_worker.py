"""One rank of the two-ranks-on-one-GPU test of the training step (started by tests/test_train_gpu.py).

Each rank runs the REAL kernels (forward, loss, backward, gradient norm, AdamW) on its own batch; the flat gradient is
summed over ranks through `gloo` (RCCL refuses two ranks on one device) exactly where `RetrieverTrainer` would call RCCL.

usage: train_rank_worker.py RANK WORLD PORT OUT_DIR
"""
import datetime
import os
import sys

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO_ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

D = H = 32
STEPS = 3


def make_model(dev):
    from evi_rag_amd.retriever import Retriever

    torch.manual_seed(3)
    return Retriever(emb_dim=D, hidden_dim=H, dropout_p=0.0, hide_seek_cfg={"enabled": False}).to(dev)


def make_batch(rank, dev):
    from evi_rag_amd import synthetic

    sb = synthetic.make_batch(4, nodes_per_graph=60, edges_per_graph=200, emb_dim=D, num_relations=12, seed=100 + rank)
    b = synthetic.as_namespace(sb, device=dev)
    b.num_relations = 12
    return b


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    dev = torch.device("cuda:0")
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.train import RetrieverTrainer

    model = make_model(dev)
    trainer = RetrieverTrainer(model, loss=RetrieverLoss(infonce_temperature=0.5), optimizer_cfg={"type": "adamw", "lr": 1e-2, "weight_decay": 1e-4},
                               scheduler_cfg={"type": "cosine", "t_max": 4, "eta_min": 1e-6}, gradient_clip_val=1.0)
    batch = make_batch(rank, dev)
    losses = []
    for _ in range(STEPS):
        losses.append(trainer.training_step(batch))
    log = trainer.on_train_epoch_end()
    # fit() over loaders of different lengths (rank 0: 3 batches, rank 1: 2): every rank stops after the common 2
    fit_log = trainer.fit([batch] * (3 - rank), max_epochs=1)
    assert fit_log["steps"] == 2, fit_log["steps"]
    out = {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}
    out["__losses"] = torch.stack(losses).cpu().numpy()
    out["__epoch_loss"] = np.array(log["train/loss"])
    out["__lr"] = np.array(log["lr"])
    np.savez(os.path.join(out_dir, f"train_rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
