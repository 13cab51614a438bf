"""`python -m evi_rag_amd.eval experiment=eval_retriever dataset=webqsp ckpt.retriever=/path/retriever.ckpt`

The reference's evaluation entry point (src/eval.py + configs/eval.yaml) for the retriever, on the MI355X backend and
without Hydra / Lightning: the SAME `configs/` directory is composed with the same command-line overrides
(`hydra_lite.compose_raw`), `model.retriever`, `model.loss` and the callbacks are built from their `_target_`s (mapped to
the mirrors in this package), the checkpoint is loaded strictly (src/eval.py:80-111), every (dataset variant, split)
the run asks for is evaluated with `RetrieverEvaluator` (the test loop of RetrieverModule) over the packed split
resident in HBM, and `metrics[_<variant>][_<split>].json` lands in `${paths.output_dir}` under the reference's naming
rules (src/eval.py:381-393).

The config directory is the reference checkout's own `configs/`: `--config-dir DIR` (or `-cd DIR`, Hydra's flag), else
$EVI_RAG_CONFIG_DIR.  Where the reference reads `<dataset.paths.embeddings>/<split>.lmdb`, this backend reads the
packed split `<dataset.paths.embeddings>/<split>.packed/` written by `evi_rag_amd.packed_dataset.write_packed` from the
same sample dictionaries (`+data.packed_root=DIR` moves it); the embedding tables are the reference's
`entity_embeddings.pt` / `relation_embeddings.pt` in the same directory.
"""
from __future__ import annotations

import copy
import json
import logging
import os
import sys
from pathlib import Path
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple

from . import hydra_lite as hl

log = logging.getLogger("evi_rag_amd.eval")

_RUNS_NEEDING_CKPT = {"eval_retriever": "retriever"}  # src/eval.py:24-28 (the GFlowNet runs are not this backend's)
_EXAMPLE = "python -m evi_rag_amd.eval experiment=eval_retriever dataset=webqsp ckpt.retriever=/path/to/retriever.ckpt"


# ---- checks the reference performs before it evaluates ---------------------------------------------------------------
def preflight_validate(cfg: Mapping[str, Any]) -> None:
    """src/eval.py:233-272: missing groups are reported as such, not as interpolation errors."""
    if cfg.get("dataset") is None:
        raise ValueError(f"Missing required config group: `dataset`.\nFix:\n  {_EXAMPLE}")
    run = cfg.get("run") or {}
    name = str(run.get("name") or "").strip()
    if name in ("", "null", "None"):
        raise ValueError(f"Missing required config group: `run`.\nFix:\n  {_EXAMPLE}")
    kind = _RUNS_NEEDING_CKPT.get(name)
    if kind and cfg.get("ckpt_path") in (None, ""):
        raise ValueError(f"Run `{name}` requires `{kind}` checkpoint, but `ckpt_path` is empty.\n"
                         f"Fix: pass `ckpt.{kind}=/path/to/{kind}.ckpt`.")
    if bool(run.get("require_dual_datasets", False)) and not run.get("dataset_variants"):
        raise ValueError("run.require_dual_datasets=true but run.dataset_variants is empty. "
                         "Provide both full and sub dataset names.")


def enforce_single_gpu_eval(trainer_cfg: Mapping[str, Any]) -> None:
    """src/eval.py:32-77: evaluation runs on exactly one GPU, no distributed strategy (sample counts and metric
    aggregation must not be sharded)."""
    accelerator = str(trainer_cfg.get("accelerator", "")).lower()
    if accelerator not in ("gpu", "cuda"):
        raise ValueError(f"Eval requires a GPU accelerator, got trainer.accelerator={trainer_cfg.get('accelerator')!r}. "
                         "Fix: set `trainer.accelerator=gpu` (and keep `trainer.devices=1`).")
    devices = trainer_cfg.get("devices")
    count: Optional[int] = None
    if isinstance(devices, bool):
        count = None
    elif isinstance(devices, int):
        count = devices
    elif isinstance(devices, (list, tuple)):
        count = len(devices)
    elif isinstance(devices, str):
        text = devices.strip().lower()
        if text.isdigit():
            count = int(text)
        elif "," in text:
            count = len([p for p in text.split(",") if p.strip()])
    if count != 1:
        raise ValueError(f"Eval forbids multi-GPU / automatic device selection, got trainer.devices={devices!r}. "
                         "Fix: set `trainer.devices=1` (pick the GPU with HIP_VISIBLE_DEVICES).")
    strategy = str(trainer_cfg.get("strategy", "auto")).lower()
    if any(tag in strategy for tag in ("ddp", "fsdp", "deepspeed")):
        raise ValueError(f"Eval forbids distributed strategies, got trainer.strategy={trainer_cfg.get('strategy')!r}. "
                         "Fix: remove the override or set `trainer.strategy=auto`.")


def dataset_scope(dataset_cfg: Mapping[str, Any]) -> str:
    scope = str(dataset_cfg.get("dataset_scope") or "").strip().lower()
    if scope in ("full", "sub"):
        return scope
    return "sub" if str(dataset_cfg.get("name") or "").endswith("-sub") else "full"


# ---- checkpoint ------------------------------------------------------------------------------------------------------
def load_checkpoint_strict(model, ckpt_path: Optional[str]) -> None:
    """The retriever's weights out of a reference checkpoint, strictly (src/eval.py:80-111).  A Lightning `.ckpt`
    stores `RetrieverModule.state_dict()`: the retriever sits under `model.` (retriever_module.py:59), torch.compile
    adds `_orig_mod.`.  Loaded with `weights_only=True`; a checkpoint that needs unpickling is refused unless
    EVI_RAG_TRUST_CKPT=1 says the file is trusted (the reference always unpickles)."""
    import torch

    if ckpt_path in (None, ""):
        return
    path = Path(str(ckpt_path))
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    try:
        blob = torch.load(str(path), map_location="cpu", weights_only=True)
    except Exception as exc:  # noqa: BLE001 - torch raises several types for unsafe pickles
        if os.environ.get("EVI_RAG_TRUST_CKPT", "") != "1":
            raise RuntimeError(f"{path} cannot be loaded with weights_only=True ({exc}); set EVI_RAG_TRUST_CKPT=1 if the "
                               "file is trusted") from exc
        blob = torch.load(str(path), map_location="cpu", weights_only=False)
    state = blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob
    if not isinstance(state, dict):
        raise TypeError(f"Checkpoint at {path} must be a state_dict mapping, got {type(state)!r}")
    cleaned: Dict[str, Any] = {}
    foreign: List[str] = []
    for key, value in state.items():
        key = key.replace("_orig_mod.", "")
        if key.startswith("model."):
            cleaned[key[len("model."):]] = value
        elif key.split(".", 1)[0] in ("loss", "train_metrics", "val_metrics", "test_metrics"):
            continue  # RetrieverModule members outside the retriever
        else:
            foreign.append(key)
    if not cleaned and foreign:  # a bare Retriever.state_dict()
        cleaned, foreign = {k.replace("_orig_mod.", ""): v for k, v in state.items()}, []
    if foreign:
        raise RuntimeError(f"Checkpoint {path} holds keys outside `model.`: {foreign[:5]}{' ...' if len(foreign) > 5 else ''}")
    model.load_state_dict(cleaned, strict=True)


# ---- one (variant, split) --------------------------------------------------------------------------------------------
def _build_callbacks(cfg: Mapping[str, Any], dataset) -> List[Any]:
    out: List[Any] = []
    for name, node in (cfg.get("callbacks") or {}).items():
        if not isinstance(node, Mapping) or "_target_" not in node:
            continue
        target = str(node["_target_"])
        if target.endswith("RetrieverTopKEdgeWriter"):
            out.append(hl.instantiate(node))
        elif target.endswith("GAgentMaterializationCallback"):
            settings = hl.instantiate(node.get("settings") or {})
            if getattr(settings, "enabled", False):
                from .g_agent import GAgentBuilder

                out.append(GAgentBuilder(settings, embedding_store=dataset))
        else:
            log.info("callback %s (%s) has no counterpart in this backend: skipped", name, target)
    return out


def evaluate(cfg: Mapping[str, Any], *, device: Optional[str] = None) -> Tuple[Dict[str, float], Dict[str, Any]]:
    """One split of one dataset: the reference's `evaluate` (src/eval.py:311-395)."""
    import torch

    from .embedding_store import GlobalEmbeddingStore
    from .eval_loop import RetrieverEvaluator
    from .packed_dataset import PackedLoader, PackedRetrievalDataset

    run = cfg.get("run")
    if run is None:
        raise ValueError(f"Missing required config group: `run`. Example: `{_EXAMPLE}`.")
    mode = str(run.get("eval_mode") or "predict").strip().lower()
    if mode not in ("predict", "test"):
        raise ValueError("run.eval_mode must be one of {'predict', 'test'}.")
    enforce_single_gpu_eval(cfg.get("trainer") or {})
    if cfg.get("seed") is not None:
        torch.manual_seed(int(cfg["seed"]))
    split = str(run.get("split", "test"))
    dev = torch.device(device or "cuda:0")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)  # the evi_* kernels launch on the current device's stream

    data_cfg = cfg.get("data") or {}
    ds_cfg = cfg["dataset"]
    split_name = str((data_cfg.get("splits") or {}).get("test", split)) if mode == "test" else split
    emb_dir = Path(str(ds_cfg["paths"]["embeddings"]))
    packed_root = Path(str(data_cfg.get("packed_root") or emb_dir))
    store = GlobalEmbeddingStore(emb_dir, device=dev)
    dataset = PackedRetrievalDataset(packed_root / f"{split_name}.packed", device=dev, embeddings=store)
    loader = PackedLoader(dataset, batch_size=int(data_cfg.get("batch_size", 32)), drop_last=bool(data_cfg.get("drop_last", False)))

    model_cfg = cfg.get("model") or {}
    model = hl.instantiate(model_cfg["retriever"]).to(dev).eval()
    loss = hl.instantiate(model_cfg["loss"]) if isinstance(model_cfg.get("loss"), Mapping) else None
    load_checkpoint_strict(model, cfg.get("ckpt_path"))
    ev_cfg = model_cfg.get("evaluation_cfg") or {}
    callbacks = _build_callbacks(cfg, dataset)
    evaluator = RetrieverEvaluator(
        model, loss=loss, k_values=list(ev_cfg.get("edge_recall_k") or (1, 10, 25, 50, 100, 200, 300, 400, 500)),
        split="test", bridge_metrics=bool(ev_cfg.get("bridge_metrics", False)),
        feature_metrics=bool(ev_cfg.get("feature_metrics", False)), ablate_topic=bool(ev_cfg.get("ablate_topic", False)),
        callbacks=callbacks, emit_predict_outputs=bool(ev_cfg.get("emit_predict_outputs", False)))
    with torch.no_grad():
        result = evaluator.run(loader)
    metrics = dict(result["metrics"])

    if metrics:
        filename = "metrics.json"
        variant = run.get("dataset_variant")
        if variant:
            filename = f"metrics_{variant}.json"
        if bool(run.get("run_all_splits", False)) and split not in (None, ""):
            filename = (f"metrics_{variant}_" if variant else "metrics_") + f"{split}.json"
        out_dir = Path(str(cfg["paths"]["output_dir"]))
        out_dir.mkdir(parents=True, exist_ok=True)
        (out_dir / filename).write_text(json.dumps(metrics, indent=2))
        log.info("Metrics saved to %s", out_dir / filename)
    else:
        log.warning("No metrics were produced; skipping metrics.json.")
    return metrics, {"cfg": cfg, "model": model, "dataset": dataset, "callbacks": callbacks,
                     "num_graphs": result["num_graphs"], "queries_per_sec": result["queries_per_sec"]}


# ---- the loops of the entry point -------------------------------------------------------------------------------------
def _variants(raw: Mapping[str, Any], hydra_node: Dict[str, Any], config_dir: Path) -> List[Tuple[str, Dict[str, Any]]]:
    """run.dataset_variants -> [(label, unresolved dataset config)] (src/eval.py:151-174)."""
    resolved_run = hl.resolve_config(raw, hydra_node).get("run") or {}
    items = resolved_run.get("dataset_variants")
    if not items:
        return []
    if not isinstance(items, (list, tuple)):
        items = [items]
    out = []
    for item in items:
        if isinstance(item, Mapping):
            name = str(item.get("dataset") or item.get("name") or "").strip()
            label = str(item.get("label") or name).strip()
        else:
            name = label = str(item).strip()
        if not name:
            raise ValueError("dataset_variants entries must define a dataset name.")
        path = config_dir / "dataset" / f"{name}.yaml"
        if not path.exists():
            raise FileNotFoundError(f"Dataset config not found: {path}")
        body, _ = hl._load(path)
        out.append((label, body))
    return out


def run(config_dir: os.PathLike, overrides: Sequence[str], *, device: Optional[str] = None,
        hydra_runtime: Optional[Mapping[str, Any]] = None) -> List[Tuple[Optional[str], str, Dict[str, float]]]:
    """Everything `main` does; returns [(dataset variant or None, split, metrics)]."""
    config_dir = Path(config_dir)
    raw, hydra_node = hl.compose_raw(config_dir, "eval", overrides, hydra_runtime=hydra_runtime)
    if raw.get("dataset") is None:
        preflight_validate({"dataset": None})
    preflight_validate(hl.resolve_config(raw, hydra_node))
    results: List[Tuple[Optional[str], str, Dict[str, float]]] = []

    def one(tree: Dict[str, Any], variant: Optional[str], split: Optional[str]) -> None:
        tree = copy.deepcopy(tree)
        run_node = tree.setdefault("run", {})
        if split is not None:
            run_node["split"] = split
        if run_node.get("allow_empty_answer") is None:
            run_node["allow_empty_answer"] = str(run_node.get("split", "test")) != "train"
        cfg = hl.resolve_config(tree, hydra_node)
        log.info("eval: dataset_variant=%s split=%s", variant, cfg["run"].get("split"))
        metrics, _ = evaluate(cfg, device=device)
        results.append((variant, str(cfg["run"].get("split")), metrics))

    def splits_of(tree: Mapping[str, Any]) -> List[Optional[str]]:
        run_node = hl.resolve_config(tree, hydra_node).get("run") or {}
        if not bool(run_node.get("run_all_splits", False)):
            return [None]
        names = [str(s) for s in (run_node.get("splits") or ["train", "validation", "test"])]
        if not names:
            raise ValueError("run.splits must be a non-empty list when run.run_all_splits=true.")
        return names

    variants = _variants(raw, hydra_node, config_dir)
    if variants:
        run_node = hl.resolve_config(raw, hydra_node).get("run") or {}
        if bool(run_node.get("require_dual_datasets", False)):
            probe = copy.deepcopy(raw)
            scopes = set()
            for _, body in variants:
                probe["dataset"] = body
                scopes.add(dataset_scope(hl.resolve_config(probe, hydra_node)["dataset"]))
            if scopes != {"full", "sub"}:
                raise ValueError("Dual-dataset evaluation requires both full and sub scopes. "
                                 f"Got scopes={sorted(scopes)} for variants={[label for label, _ in variants]}.")
        for label, body in variants:
            tree = copy.deepcopy(raw)
            tree["dataset"] = body
            tree.setdefault("run", {})["dataset_variant"] = label
            for split in splits_of(tree):
                one(tree, label, split)
        return results
    for split in splits_of(raw):
        one(raw, None, split)
    return results


def main(argv: Optional[Sequence[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    config_dir = os.environ.get("EVI_RAG_CONFIG_DIR")
    overrides: List[str] = []
    i = 0
    while i < len(argv):
        arg = argv[i]
        if arg in ("--config-dir", "-cd", "--config-path", "-cp"):
            if i + 1 >= len(argv):
                raise SystemExit(f"{arg} needs a directory")
            config_dir = argv[i + 1]
            i += 2
            continue
        if arg.startswith("--config-dir="):
            config_dir = arg.split("=", 1)[1]
        elif arg.startswith("-") and "=" not in arg:
            raise SystemExit(f"unknown flag {arg!r}; usage: {_EXAMPLE} --config-dir /path/to/EVI-RAG/configs")
        else:
            overrides.append(arg)
        i += 1
    if not config_dir:
        raise SystemExit("the reference's configs/ directory is required: --config-dir DIR or $EVI_RAG_CONFIG_DIR")
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")
    for variant, split, metrics in run(config_dir, overrides):
        print(json.dumps({"dataset_variant": variant, "split": split, "metrics": metrics}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
