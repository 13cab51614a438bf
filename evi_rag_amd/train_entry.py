"""`python -m evi_rag_amd.train_entry --config-dir DIR experiment=train_retriever dataset=webqsp [overrides]` — the retriever
run of the reference's `src/train.py` + `configs/train.yaml` on this backend, without Hydra / Lightning.

What the reference's run does for the retriever (src/train.py: instantiate datamodule / model / callbacks / trainer, `trainer.fit`),
reduced to what changes numbers or files:
  * config: the reference's own `configs/` tree composed by `hydra_lite` (`train.yaml` defaults, `experiment=train_retriever`);
  * data: `<dataset.paths.embeddings>/<split>.packed` (see `evi_rag_amd.eval`), `data.splits.{train,validation}`, `data.batch_size`,
    shuffled with `seed`; one process per GPU shares the split by rank;
  * model / loss: `model.retriever`, `model.loss` through `hydra_lite.instantiate` (mapped to the mirrors);
  * optimiser / schedule / clipping: `model.optimizer_cfg`, `model.scheduler_cfg`, `trainer.gradient_clip_val` -> `RetrieverTrainer`;
  * `trainer.precision`: `bf16-mixed` -> single-product bf16 GEMMs in the scorer (forward, backward and the validation inside the
    run); `32-true` and the default `16-mixed` -> the split-bf16 (f32-grade) GEMMs (`train.matmul_precision_for`);
  * validation every `trainer.check_val_every_n_epoch` epochs with `RetrieverEvaluator` (metrics under `val/`, summed over ranks);
  * `callbacks.model_checkpoint` (`dirpath`, `filename` with `{epoch:03d}`, `monitor`, `mode`, `save_last`): the best checkpoint by the
    monitored validation metric and `last.ckpt`, in Lightning's layout as far as `src/eval.py` reads it;
  * `callbacks.early_stopping` (`monitor`, `mode`, `patience`, `min_delta`); `trainer.min_epochs` / `max_epochs`; `ckpt_path` resumes.
Loggers, progress bars, model summary, `test: True` after training (run `evi_rag_amd.eval` on the checkpoint) are not mirrored.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
from pathlib import Path
from typing import Any, Dict, Mapping, Optional, Sequence

from . import hydra_lite as hl

log = logging.getLogger("evi_rag_amd.train_entry")


def _better(value: float, best: Optional[float], mode: str, min_delta: float = 0.0) -> bool:
    if best is None:
        return True
    return value > best + min_delta if mode == "max" else value < best - min_delta


def fit(cfg: Mapping[str, Any], *, device: Optional[str] = None) -> Dict[str, Any]:
    import torch
    import torch.distributed as dist

    from .embedding_store import GlobalEmbeddingStore
    from .eval_loop import RetrieverEvaluator
    from .packed_dataset import PackedLoader, PackedRetrievalDataset
    from .train import RetrieverTrainer

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    dev = torch.device(device or f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)  # the evi_* kernels launch on the CURRENT device's stream: make `--device cuda:1` that device
    seed = cfg.get("seed")
    if seed is not None:
        torch.manual_seed(int(seed))  # every rank: the same initial weights and the same shuffling stream (shares are disjoint)

    data_cfg = cfg.get("data") or {}
    ds_cfg = cfg.get("dataset")
    if ds_cfg is None:
        raise ValueError("Missing required config group: `dataset`. Example: `experiment=train_retriever dataset=webqsp`.")
    splits = data_cfg.get("splits") or {}
    emb_dir = Path(str(ds_cfg["paths"]["embeddings"]))
    packed_root = Path(str(data_cfg.get("packed_root") or emb_dir))
    store = GlobalEmbeddingStore(emb_dir, device=dev)
    bs = int(data_cfg.get("batch_size", 32))
    train_ds = PackedRetrievalDataset(packed_root / f"{splits.get('train', 'train')}.packed", device=dev, embeddings=store)
    val_path = packed_root / f"{splits.get('validation', 'validation')}.packed"
    val_ds = PackedRetrievalDataset(val_path, device=dev, embeddings=store) if val_path.exists() else None
    train_loader = PackedLoader(train_ds, batch_size=bs, shuffle=True, random_seed=int(seed) if seed is not None else None,
                                drop_last=bool(data_cfg.get("drop_last", False)), rank=rank, world_size=world)

    model_cfg = cfg.get("model") or {}
    model = hl.instantiate(model_cfg["retriever"]).to(dev)
    loss = hl.instantiate(model_cfg["loss"]) if isinstance(model_cfg.get("loss"), Mapping) else None
    tr_cfg = cfg.get("trainer") or {}
    trainer = RetrieverTrainer(model, loss=loss, optimizer_cfg=model_cfg.get("optimizer_cfg"), scheduler_cfg=model_cfg.get("scheduler_cfg"),
                               gradient_clip_val=tr_cfg.get("gradient_clip_val"), precision=tr_cfg.get("precision"))
    resumed: Dict[str, Any] = {}
    if cfg.get("ckpt_path"):
        resumed = trainer.load_checkpoint(cfg["ckpt_path"]) or {}
        log.info("resumed from %s at epoch %d", cfg["ckpt_path"], trainer.current_epoch)

    cbs = cfg.get("callbacks") or {}
    mc = cbs.get("model_checkpoint") or {}
    es = cbs.get("early_stopping") or {}
    ckpt_dir = Path(str(mc.get("dirpath") or Path(str(cfg["paths"]["output_dir"])) / "checkpoints"))
    monitor, mode = mc.get("monitor"), str(mc.get("mode", "min"))
    es_monitor, es_mode = es.get("monitor"), str(es.get("mode", "min"))
    patience, min_delta = int(es.get("patience", 3)), float(es.get("min_delta", 0.0))
    ev_cfg = model_cfg.get("evaluation_cfg") or {}
    k_values = list(ev_cfg.get("edge_recall_k") or (1, 10, 25, 50, 100, 200, 300, 400, 500))
    every = max(int(tr_cfg.get("check_val_every_n_epoch", 1) or 1), 1)
    min_epochs, max_epochs = int(tr_cfg.get("min_epochs", 0) or 0), int(tr_cfg.get("max_epochs", 1) or 1)

    # selection state of model_checkpoint / early_stopping: restored from the checkpoint a run resumes from (Lightning keeps it
    # in the checkpoint's `callbacks` entry), so the first validation after a resume competes with the best one so far
    history = []
    best = resumed.get("best")
    best_path = Path(str(resumed["best_path"])) if resumed.get("best_path") else None
    es_best, bad_checks = resumed.get("es_best"), int(resumed.get("bad_checks") or 0)
    stop_latched = bool(resumed.get("stop_latched", False))

    def cb_state():
        return {"best": best, "best_path": str(best_path) if best_path is not None else None, "es_best": es_best,
                "bad_checks": bad_checks, "stop_latched": stop_latched}

    if rank == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    while trainer.current_epoch < max_epochs:
        epoch = trainer.current_epoch
        if stop_latched and epoch >= min_epochs:
            log.info("early stopping: patience was exhausted before the resume point")
            break
        if seed is not None:
            # the dropout-seed stream (drawn from torch's CPU generator per step) is a function of (seed, epoch) too: epoch e of a
            # resumed run draws what epoch e of the uninterrupted run drew
            torch.manual_seed(int(seed) * 1_000_003 + epoch)
        entry = dict(trainer.fit(train_loader, max_epochs=1)["epochs"][-1], epoch=epoch)
        stop = False
        if val_ds is not None and (epoch + 1) % every == 0:
            model.eval()
            evaluator = RetrieverEvaluator(model, loss=loss, k_values=k_values, split="val", bridge_metrics=bool(ev_cfg.get("bridge_metrics", False)))
            with torch.no_grad():
                res = evaluator.run(PackedLoader(val_ds, batch_size=bs, rank=rank, world_size=world), sync=world > 1)
            entry.update(res["metrics"])
            # early stopping first, so that the checkpoint written below carries this check's counters
            if es_monitor is not None and es_monitor in res["metrics"]:
                if _better(float(res["metrics"][es_monitor]), es_best, es_mode, min_delta):
                    es_best, bad_checks = float(res["metrics"][es_monitor]), 0
                else:
                    bad_checks += 1
                    stop_latched = stop_latched or bad_checks >= patience  # Lightning latches should_stop ...
            if monitor is not None:
                if monitor not in res["metrics"]:
                    raise KeyError(f"model_checkpoint.monitor {monitor!r} is not among the validation metrics {sorted(res['metrics'])}")
                if _better(float(res["metrics"][monitor]), best, mode):
                    best = float(res["metrics"][monitor])
                    name = str(mc.get("filename") or "epoch_{epoch:03d}").format(epoch=epoch) + ".ckpt"
                    old_path, best_path = best_path, ckpt_dir / name
                    if rank == 0:
                        if old_path is not None and old_path != best_path and old_path.exists() and int(mc.get("save_top_k", 1) or 1) == 1:
                            old_path.unlink()
                        trainer.save_checkpoint(best_path, callbacks=cb_state())
        stop = stop_latched and (epoch + 1) >= min_epochs  # ... and acts on it as soon as min_epochs allows, improving check or not
        if rank == 0 and bool(mc.get("save_last", False)):
            trainer.save_checkpoint(ckpt_dir / "last.ckpt", callbacks=cb_state())
        history.append(entry)
        log.info("epoch %d: %s", epoch, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in entry.items()})
        if stop:
            log.info("early stopping at epoch %d (%s has not improved for %d checks)", epoch, es_monitor, bad_checks)
            break
    if rank == 0:
        out_dir = Path(str(cfg["paths"]["output_dir"]))
        out_dir.mkdir(parents=True, exist_ok=True)
        (out_dir / "train_history.json").write_text(json.dumps(history, indent=2))
    return {"history": history, "best": best, "best_checkpoint": str(best_path) if best_path is not None else None,
            "checkpoint_dir": str(ckpt_dir), "trainer": trainer}


def run(config_dir: os.PathLike, overrides: Sequence[str], *, device: Optional[str] = None, searchpath: Sequence[str] = ()) -> Dict[str, Any]:
    """Compose `train.yaml` of `config_dir` with the command-line overrides and train."""
    raw, hydra_node = hl.compose_raw(Path(config_dir), "train", list(overrides), searchpath=list(searchpath))
    cfg = hl.resolve_config(raw, hydra_node)
    if not cfg.get("train", True):
        raise ValueError("train=False: nothing to do (evaluation of a checkpoint is `python -m evi_rag_amd.eval`).")
    return fit(cfg, device=device)


def main(argv: Optional[Sequence[str]] = None) -> int:
    import torch
    import torch.distributed as dist

    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config-dir", required=True, help="the reference's configs/ directory (or a tree of the same shape)")
    ap.add_argument("--searchpath", action="append", default=[], help="extra config roots (the overlay in this repo's configs/)")
    ap.add_argument("--device", default=None)
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        if os.environ.get("EVI_NCCL_EAGER_INIT", "") == "1":
            dist.init_process_group("nccl", device_id=dev)  # eager init, sub-groups by ncclCommSplit
        else:
            dist.init_process_group("nccl")  # communicators built by their first collective (the device is current: set_device above)
    try:
        out = run(args.config_dir, args.overrides, device=args.device, searchpath=args.searchpath)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"epochs": len(out["history"]), "best": out["best"], "best_checkpoint": out["best_checkpoint"]}))
    finally:
        if world > 1:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
