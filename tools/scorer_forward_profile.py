#!/usr/bin/env python3
"""Runs the WebQSP-shaped scorer forward (32 graphs, E ~ 131k, D = H = 768) a few times: the workload for
`rocprofv3 --kernel-trace --stats -- python3 tools/scorer_forward_profile.py [lite]` (per-kernel time of the forward)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from evi_rag_amd import synthetic
from evi_rag_amd.retriever import Retriever


def main():
    dev = torch.device("cuda:0")
    D = int(os.environ.get("EVI_PROFILE_D", "768"))
    sb = synthetic.make_batch(32, nodes_per_graph=1500, edges_per_graph=4096, emb_dim=D, num_relations=4096, num_entities=1 << 17, seed=1)
    batch = synthetic.as_namespace(sb, device=dev)
    batch.num_relations = 4096
    torch.manual_seed(0)
    model = Retriever(emb_dim=D, hidden_dim=D).to(dev).eval()
    model.emit_edge_embeddings = not (len(sys.argv) > 1 and sys.argv[1] in ("lite", "bwd"))
    if len(sys.argv) > 1 and sys.argv[1] == "bwd":  # forward + backward (evi_retriever_backward recomputes the forward inside)
        model.differentiable = True
        g = torch.randn(sb.num_edges, device=dev)
        for _ in range(2):
            model.zero_grad()
            (model(batch).logits * g).sum().backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 5
        for _ in range(n):
            model.zero_grad()
            (model(batch).logits * g).sum().backward()
        e1.record()
        torch.cuda.synchronize()
        print(f"forward + backward: {e0.elapsed_time(e1) / n:.3f} ms per batch, E={sb.num_edges} N={sb.num_nodes}")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "train":  # the whole optimiser step (train.RetrieverTrainer), reference defaults
        import time

        from evi_rag_amd.loss import RetrieverLoss
        from evi_rag_amd.train import RetrieverTrainer

        tm = Retriever(emb_dim=D, hidden_dim=D, dropout_p=float(os.environ.get("EVI_PROFILE_DROPOUT", "0.1")),
                       hide_seek_cfg={"enabled": os.environ.get("EVI_PROFILE_HIDE_SEEK", "1") == "1", "p_near": 0.7, "p_far": 0.1,
                                      "bias_near": -2.0, "bias_far": -0.5}).to(dev)
        tm.emit_edge_embeddings = False
        tr = RetrieverTrainer(tm, loss=RetrieverLoss(infonce_temperature=0.07), precision=os.environ.get("EVI_PROFILE_PRECISION", "32-true"))
        for _ in range(2):
            tr.training_step(batch)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            tr.training_step(batch)
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"training step: {e0.elapsed_time(e1) / n:.3f} ms per batch on the device clock, host issue time {t_host / n * 1e3:.3f} ms, "
              f"E={sb.num_edges} N={sb.num_nodes}")
        return
    for _ in range(3):
        model(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        model(batch)
    e1.record()
    torch.cuda.synchronize()
    print(f"forward ({'logits only' if not model.emit_edge_embeddings else 'with edge features'}): {e0.elapsed_time(e1) / n:.3f} ms per batch, "
          f"E={sb.num_edges} N={sb.num_nodes}")


if __name__ == "__main__":
    main()
