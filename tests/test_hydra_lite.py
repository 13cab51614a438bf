"""CPU: composition of a reference-shaped config tree (hydra_lite) and the pure checks of the eval entry point."""
from pathlib import Path

import pytest
import torch

from tests.config_tree import write_tree

REFERENCE_CONFIGS = Path("/root/reference/configs")


def test_compose_experiment_overlay_and_interpolation(tmp_path, monkeypatch):
    from evi_rag_amd import hydra_lite as hl

    monkeypatch.setenv("EVI_TEST_PROJECT_ROOT", "/proj")
    monkeypatch.delenv("EVI_TEST_RETRIEVER_CKPT", raising=False)
    cfg_dir = write_tree(tmp_path, tmp_path / "data")
    cfg = hl.compose(cfg_dir, "eval", ["experiment=eval_retriever", "dataset=toyqa", "ckpt.retriever=/w/r.ckpt", "+run.extra=[1, 2]"])
    # primary body first (`_self_` leads), groups after, the experiment overlay (global package) last
    assert cfg["task_name"] == "eval/retriever" and cfg["tags"] == ["eval", "retriever"] and cfg["seed"] == 7
    assert cfg["ckpt_path"] == "/w/r.ckpt" == cfg["run"]["ckpt_path"]  # ${run.ckpt_path} -> ${ckpt.retriever} -> CLI value
    assert cfg["run"]["extra"] == [1, 2] and cfg["run"]["run_all_splits"] is True and cfg["run"]["splits"] == ["validation", "test"]
    # `override /model: retriever_module` selected a group the primary lists as null; typed whole-string interpolation
    r = cfg["model"]["retriever"]
    assert r["num_topics"] == 2 and r["dde_cfg"] == {"num_rounds": 2, "num_reverse_rounds": 2} and r["emb_dim"] == 16
    assert cfg["model"]["compile_model"] is False and cfg["model"]["evaluation_cfg"]["edge_recall_k"] == [1, 5, 20]
    assert cfg["model"]["evaluation_cfg"]["split"] == "test"
    # sibling defaults inside the callbacks group land in the group's package; oc.select with and without the key
    g = cfg["callbacks"]["g_agent_materializer"]["settings"]
    assert g["enabled"] is False and g["edge_top_k"] == 30 and g["start_keep_ratio"] == 0.25 and g["allow_empty_answer"] is False
    w = cfg["callbacks"]["retriever_topk_edge_writer"]
    assert w["output_dir"] == f"{tmp_path}/data/toyqa/artifacts/toyqa/eval_retriever" and w["topk_values"] == [1, 5, 20]
    # nested env default, the hydra run dir behind ${hydra:runtime.output_dir}, choices visible to oc.select
    assert cfg["paths"]["root_dir"] == "/proj" and cfg["paths"]["output_dir"] == "/proj/logs//eval_retriever_toyqa/runs/fixed"
    assert cfg["trainer"]["default_root_dir"] == cfg["paths"]["output_dir"] and "hydra" not in cfg
    assert cfg["run"]["dataset_variants"] == ["toyqa", "toyqa-sub"] and cfg["data"]["splits"]["test"] == "test"
    assert cfg["data"]["dataset_cfg"]["paths"]["embeddings"] == f"{tmp_path}/data/toyqa/materialized/embeddings"
    # an env default of null stays None; the optional `local` group is absent and skipped
    cfg2 = hl.compose(cfg_dir, "eval", ["experiment=eval_retriever", "dataset=toyqa-sub", "run.split=validation"])
    assert cfg2["ckpt_path"] is None and cfg2["dataset"]["dataset_scope"] == "sub" and cfg2["data"]["splits"]["test"] == "validation"
    monkeypatch.setenv("EVI_TEST_RETRIEVER_CKPT", "/env/ckpt")
    assert hl.compose(cfg_dir, "eval", ["experiment=eval_retriever", "dataset=toyqa"])["ckpt_path"] == "/env/ckpt"


def test_compose_errors_are_loud(tmp_path):
    from evi_rag_amd import hydra_lite as hl

    cfg_dir = write_tree(tmp_path, tmp_path / "data")
    with pytest.raises(hl.ConfigError, match="no option"):
        hl.compose(cfg_dir, "eval", ["dataset=nope"])
    with pytest.raises(hl.ConfigError, match="not found"):  # experiment needs ${dataset.*}: missing group -> named key
        hl.compose(cfg_dir, "eval", ["experiment=eval_retriever"])
    with pytest.raises(hl.ConfigError, match="key=value"):
        hl.compose(cfg_dir, "eval", ["dataset"])
    with pytest.raises(hl.ConfigError, match="cycle"):
        hl.resolve_all({"a": "${b}", "b": "${a}"})
    with pytest.raises(hl.ConfigError, match="unsupported resolver"):
        hl.resolve_all({"a": "${oc.decode:x}"})
    with pytest.raises(hl.ConfigError, match="not set"):
        hl.resolve_all({"a": "${oc.env:EVI_SURELY_UNSET_VARIABLE}"})
    with pytest.raises(FileNotFoundError):
        hl.compose(cfg_dir, "no_such_primary", [])
    assert hl.resolve_all({"a": {"b": [10, 20]}, "c": "x${a.b.1}y", "d": "${a.b}"}) == {"a": {"b": [10, 20]}, "c": "x20y", "d": [10, 20]}


def test_instantiate_maps_reference_targets_to_the_mirrors(tmp_path):
    from evi_rag_amd import hydra_lite as hl
    from evi_rag_amd.g_agent import GAgentSettings
    from evi_rag_amd.loss import RetrieverLoss
    from evi_rag_amd.retriever import Retriever

    cfg = hl.compose(write_tree(tmp_path, tmp_path / "data"), "eval", ["experiment=eval_retriever", "dataset=toyqa", "run.build_g_agent=true"])
    model = hl.instantiate(cfg["model"]["retriever"])
    assert isinstance(model, Retriever) and model.state_dict()["state_net.0.weight"].shape == (16, 3 * 16 + 1)
    assert isinstance(hl.instantiate(cfg["model"]["loss"]), RetrieverLoss)
    settings = hl.instantiate(cfg["callbacks"]["g_agent_materializer"]["settings"])  # nested _target_, resolved interpolations
    assert isinstance(settings, GAgentSettings) and settings.enabled and settings.edge_top_k == 30 and settings.max_hops == 3
    part = hl.instantiate({"_target_": "src.losses.retriever_loss.RetrieverLoss", "_partial_": True, "infonce_temperature": 0.5})
    assert part().infonce_temperature == 0.5
    with pytest.raises(hl.ConfigError, match="cannot locate"):
        hl.instantiate({"_target_": "src.models.retriever_module.RetrieverModule"})  # Lightning module: not a mirror


def test_eval_entry_checks_and_checkpoint_prefixes(tmp_path):
    from evi_rag_amd import eval as ev
    from evi_rag_amd.retriever import Retriever

    with pytest.raises(ValueError, match="dataset"):
        ev.preflight_validate({"dataset": None})
    with pytest.raises(ValueError, match="`run`"):
        ev.preflight_validate({"dataset": {}, "run": {"name": None}})
    with pytest.raises(ValueError, match="requires `retriever` checkpoint"):
        ev.preflight_validate({"dataset": {}, "run": {"name": "eval_retriever"}, "ckpt_path": None})
    with pytest.raises(ValueError, match="dataset_variants is empty"):
        ev.preflight_validate({"dataset": {}, "run": {"name": "x", "require_dual_datasets": True}})
    ev.enforce_single_gpu_eval({"accelerator": "gpu", "devices": 1})
    ev.enforce_single_gpu_eval({"accelerator": "cuda", "devices": "0,"})
    for bad in ({"accelerator": "cpu", "devices": 1}, {"accelerator": "gpu", "devices": 2}, {"accelerator": "gpu", "devices": "auto"},
                {"accelerator": "gpu", "devices": [0, 1]}, {"accelerator": "gpu", "devices": 1, "strategy": "ddp_find_unused"}):
        with pytest.raises(ValueError):
            ev.enforce_single_gpu_eval(bad)
    assert ev.dataset_scope({"name": "webqsp-sub"}) == "sub" and ev.dataset_scope({"name": "x", "dataset_scope": "FULL"}) == "full"

    torch.manual_seed(0)
    src = Retriever(emb_dim=16, hidden_dim=16)
    state = {f"model._orig_mod.{k}": v for k, v in src.state_dict().items()}  # Lightning module + torch.compile prefixes
    torch.save({"state_dict": state, "epoch": 3}, tmp_path / "lightning.ckpt")
    dst = Retriever(emb_dim=16, hidden_dim=16)
    ev.load_checkpoint_strict(dst, str(tmp_path / "lightning.ckpt"))
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    torch.save(src.state_dict(), tmp_path / "bare.pt")
    ev.load_checkpoint_strict(Retriever(emb_dim=16, hidden_dim=16), str(tmp_path / "bare.pt"))
    torch.save({"state_dict": {**{f"model.{k}": v for k, v in src.state_dict().items()}, "other.weight": torch.zeros(1)}}, tmp_path / "odd.ckpt")
    with pytest.raises(RuntimeError, match="outside `model.`"):
        ev.load_checkpoint_strict(Retriever(emb_dim=16, hidden_dim=16), str(tmp_path / "odd.ckpt"))
    with pytest.raises(RuntimeError):  # strict: a missing tensor is an error
        bad = {f"model.{k}": v for k, v in src.state_dict().items() if not k.startswith("score_head")}
        torch.save({"state_dict": bad}, tmp_path / "short.ckpt")
        ev.load_checkpoint_strict(Retriever(emb_dim=16, hidden_dim=16), str(tmp_path / "short.ckpt"))
    with pytest.raises(FileNotFoundError):
        ev.load_checkpoint_strict(dst, str(tmp_path / "absent.ckpt"))
    with pytest.raises(SystemExit):
        ev.main(["experiment=eval_retriever"])  # no config directory given


@pytest.mark.skipif(not REFERENCE_CONFIGS.is_dir(), reason="the reference checkout is only present in the build container")
def test_compose_the_reference_tree_itself(monkeypatch):
    """The real configs/ of the reference: `experiment=eval_retriever dataset=webqsp ckpt.retriever=X` composes, and the
    values the retriever evaluation reads come out as the YAML files state them."""
    from evi_rag_amd import hydra_lite as hl

    monkeypatch.setenv("PROJECT_ROOT", "/proj")
    cfg = hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever", "dataset=webqsp", "ckpt.retriever=/w/r.ckpt"])
    r = cfg["model"]["retriever"]
    assert r["_target_"] == "src.models.components.retriever.Retriever" and r["emb_dim"] == 1024 and r["num_topics"] == 2
    assert r["dde_cfg"] == {"num_rounds": 2, "num_reverse_rounds": 2}
    assert cfg["model"]["evaluation_cfg"]["edge_recall_k"] == [1, 10, 25, 50, 100, 200, 300, 400, 500]
    assert cfg["run"]["name"] == "eval_retriever" and cfg["run"]["eval_mode"] == "test" and cfg["ckpt_path"] == "/w/r.ckpt"
    assert cfg["run"]["dataset_variants"] == ["webqsp", "webqsp-sub"] and cfg["data"]["batch_size"] == 32
    assert cfg["callbacks"]["retriever_topk_edge_writer"]["output_dir"].endswith("/webqsp/artifacts/webqsp/eval_retriever")
    assert cfg["callbacks"]["g_agent_materializer"]["settings"]["enabled"] is False
    assert cfg["trainer"]["devices"] == 1 and cfg["paths"]["root_dir"] == "/proj" and "hydra" not in cfg
    cwq = hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever", "dataset=cwq", "ckpt.retriever=/w/r.ckpt"])
    assert cwq["dataset"]["name"] == "cwq" and cwq["run"]["dataset_variants"] == ["cwq", "cwq-sub"]


@pytest.mark.skipif(not REFERENCE_CONFIGS.is_dir(), reason="the reference checkout is only present in the build container")
def test_overlay_composes_over_the_reference_tree(monkeypatch):
    """The shipped configs/ overlay over the reference's real tree, found through hydra.searchpath:
    everything equals the plain `eval_retriever` composition except the swapped `_target_`s, and those instantiate."""
    from pathlib import Path

    from evi_rag_amd import hydra_lite as hl

    overlay = Path(__file__).resolve().parent.parent / "configs"
    monkeypatch.setenv("PROJECT_ROOT", "/proj")
    base = hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever", "dataset=webqsp", "ckpt.retriever=/w/r.ckpt"])
    cfg = hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever_mi355x", "dataset=webqsp", "ckpt.retriever=/w/r.ckpt",
                                                 f"hydra.searchpath=[file://{overlay}]"])
    assert cfg["model"]["retriever"]["_target_"] == "evi_rag_amd.retriever.Retriever"
    assert cfg["model"]["loss"]["_target_"] == "evi_rag_amd.loss.RetrieverLoss"
    assert cfg["callbacks"]["retriever_topk_edge_writer"]["_target_"] == "evi_rag_amd.topk_writer.RetrieverTopKEdgeWriter"
    # identical apart from the three swapped targets
    import copy

    same = copy.deepcopy(cfg)
    same["model"]["retriever"]["_target_"] = base["model"]["retriever"]["_target_"]
    same["model"]["loss"]["_target_"] = base["model"]["loss"]["_target_"]
    same["callbacks"]["retriever_topk_edge_writer"]["_target_"] = base["callbacks"]["retriever_topk_edge_writer"]["_target_"]
    import json
    import re

    def norm(c):  # the run directory carries the experiment's name and a timestamp
        return re.sub(r"\d{4}-\d{2}-\d{2}_\d{2}-\d{2}-\d{2}", "T", json.dumps(c, sort_keys=True).replace("eval_retriever_mi355x", "eval_retriever"))

    assert norm(same) == norm(base)
    # the same through the keyword instead of the command-line override
    assert norm(hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever_mi355x", "dataset=webqsp", "ckpt.retriever=/w/r.ckpt"],
                           searchpath=[overlay])) == norm(cfg)
    # the group-level overlays
    m = hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever", "model=retriever_module_mi355x", "dataset=webqsp",
                                               "ckpt.retriever=/w/r.ckpt"], searchpath=[overlay])
    assert m["model"]["retriever"]["_target_"] == "evi_rag_amd.retriever.Retriever" and m["model"]["retriever"]["emb_dim"] == 1024
    assert m["model"]["loss"]["infonce_temperature"] == 0.07
    # and the swapped targets instantiate (host-side construction only: no GPU call)
    model = hl.instantiate(cfg["model"]["retriever"])
    from evi_rag_amd.retriever import Retriever

    assert isinstance(model, Retriever) and model.emb_dim == 1024
    with pytest.raises(hl.ConfigError, match="pkg://"):
        hl.compose(REFERENCE_CONFIGS, "eval", ["experiment=eval_retriever", "hydra.searchpath=[pkg://x]"])


def test_training_tree_composes(tmp_path, monkeypatch):
    """`train.yaml` + `experiment=train_retriever` of the miniature tree (same shape as the reference's): the values the training
    entry point reads — optimiser, schedule, clipping, checkpoint / early-stopping callbacks — come out as the overlays state."""
    from evi_rag_amd import hydra_lite as hl
    from tests.config_tree import write_tree

    monkeypatch.setenv("EVI_TEST_PROJECT_ROOT", str(tmp_path / "proj"))
    cfg = hl.compose(write_tree(tmp_path, tmp_path / "data"), "train", ["experiment=train_retriever", "dataset=toyqa"])
    assert cfg["model"]["optimizer_cfg"] == {"type": "adamw", "lr": 3.0e-3, "weight_decay": 1.0e-4}
    assert cfg["model"]["scheduler_cfg"]["type"] == "cosine" and cfg["trainer"]["gradient_clip_val"] == 1.0
    assert cfg["trainer"]["max_epochs"] == 6 and cfg["trainer"]["check_val_every_n_epoch"] == 1 and cfg["trainer"]["devices"] == 1
    mc = cfg["callbacks"]["model_checkpoint"]
    assert mc["monitor"] == "val/answer/reachability@20" and mc["mode"] == "max" and mc["save_last"] is True
    assert mc["dirpath"].replace("//", "/").endswith("/logs/train_retriever_toyqa/runs/fixed/checkpoints")
    assert cfg["callbacks"]["early_stopping"]["patience"] == 10 and cfg["seed"] == 42


@pytest.mark.skipif(not REFERENCE_CONFIGS.is_dir(), reason="the reference checkout is only present in the build container")
def test_compose_the_reference_training_tree(monkeypatch):
    """The real configs/ of the reference: `train.yaml` + `experiment=train_retriever dataset=webqsp` composes, and what the
    training entry point reads comes out as the YAML files state it (configs/model/retriever_module.yaml:8-47,
    configs/trainer/default.yaml, configs/experiment/train_retriever.yaml)."""
    from evi_rag_amd import hydra_lite as hl

    monkeypatch.setenv("PROJECT_ROOT", "/proj")
    cfg = hl.compose(REFERENCE_CONFIGS, "train", ["experiment=train_retriever", "dataset=webqsp"])
    assert cfg["model"]["optimizer_cfg"] == {"type": "adamw", "lr": 1.0e-3, "weight_decay": 1.0e-4}
    assert cfg["model"]["scheduler_cfg"]["type"] == "cosine" and cfg["model"]["scheduler_cfg"]["t_max"] == 200
    assert cfg["trainer"]["gradient_clip_val"] == 1.0 and cfg["trainer"]["max_epochs"] == 10000 and cfg["trainer"]["check_val_every_n_epoch"] == 1
    r = cfg["model"]["retriever"]
    assert r["dropout_p"] == 0.1 and r["hide_seek_cfg"]["enabled"] is True and r["hide_seek_cfg"]["p_near"] == 0.7
    assert cfg["model"]["loss"]["infonce_weight"] == 1.0 and cfg["model"]["loss"]["bce_weight"] == 0.0
    mc = cfg["callbacks"]["model_checkpoint"]
    assert mc["monitor"] == "val/answer/reachability@100" and mc["mode"] == "max" and mc["save_last"] is True
    assert cfg["callbacks"]["early_stopping"]["patience"] == 10 and cfg["data"]["splits"]["train"] == "train"


@pytest.mark.skipif(not REFERENCE_CONFIGS.is_dir(), reason="the reference checkout is only present in the build container")
def test_training_overlay_composes_over_the_reference_tree(monkeypatch):
    """configs/experiment/train_retriever_mi355x.yaml over the reference's real tree: the reference's `train_retriever` with the
    two `_target_`s swapped and full precision selected; nothing else differs."""
    import copy
    import json
    import re
    from pathlib import Path

    from evi_rag_amd import hydra_lite as hl

    overlay = Path(__file__).resolve().parent.parent / "configs"
    monkeypatch.setenv("PROJECT_ROOT", "/proj")
    base = hl.compose(REFERENCE_CONFIGS, "train", ["experiment=train_retriever", "dataset=webqsp"])
    cfg = hl.compose(REFERENCE_CONFIGS, "train", ["experiment=train_retriever_mi355x", "dataset=webqsp"], searchpath=[overlay])
    assert cfg["model"]["retriever"]["_target_"] == "evi_rag_amd.retriever.Retriever" and cfg["model"]["loss"]["_target_"] == "evi_rag_amd.loss.RetrieverLoss"
    assert cfg["trainer"]["precision"] == "32-true" and cfg["model"]["retriever"]["dropout_p"] == 0.1
    same = copy.deepcopy(cfg)
    same["model"]["retriever"]["_target_"] = base["model"]["retriever"]["_target_"]
    same["model"]["loss"]["_target_"] = base["model"]["loss"]["_target_"]
    same["trainer"]["precision"] = base["trainer"]["precision"]

    def norm(c):
        return re.sub(r"\\d{4}-\\d{2}-\\d{2}_\\d{2}-\\d{2}-\\d{2}", "T", json.dumps(c, sort_keys=True).replace("train_retriever_mi355x", "train_retriever"))

    assert norm(same) == norm(base)
