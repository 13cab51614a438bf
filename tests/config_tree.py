"""A miniature config tree with the SHAPE of the reference's `configs/` (defaults lists, an experiment overlay in the
global package, `override /group` lines, sibling defaults inside a group, oc.env / oc.select / hydra interpolations),
written out programmatically so that the composition code is tested without the reference checkout."""
from pathlib import Path

import yaml


def _dump(path: Path, body, header: str = "") -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(header + yaml.safe_dump(body, sort_keys=False))


def write_tree(root: Path, data_dir: Path, *, emb_dim: int = 16, hidden_dim: int = 16) -> Path:
    cfg = Path(root) / "configs"
    _dump(cfg / "eval.yaml", {
        "defaults": ["_self_", {"window": "default"}, {"ckpt": "default"}, {"run": "default"}, {"dataset": None}, {"data": None},
                     {"model": None}, {"callbacks": "default"}, {"logger": None}, {"trainer": "predict"}, {"paths": "default"},
                     {"hydra": "default"}, {"experiment": None}, {"optional local": "default"}],
        "task_name": "${run.task_name}", "tags": "${run.tags}", "seed": 7, "ckpt_path": "${run.ckpt_path}"})
    _dump(cfg / "window" / "default.yaml", {"k_values": [1, 5, 20]})
    _dump(cfg / "ckpt" / "default.yaml", {"retriever": None, "gflownet": None})
    _dump(cfg / "run" / "default.yaml", {"name": None, "task_name": "eval", "tags": [], "split": "test", "run_all_splits": False,
                                         "splits": ["validation", "test"], "score_temperature": 1.0, "ckpt_path": None,
                                         "dataset_variants": None, "require_dual_datasets": False, "dataset_variant": None,
                                         "eval_mode": "predict"})
    for name, scope in (("toyqa", "full"), ("toyqa-sub", "sub")):
        _dump(cfg / "dataset" / f"{name}.yaml", {
            "name": name, "dataset_scope": scope, "dataset_family": "toyqa",
            "out_dir": "${paths.data_dir}/" + name + "/normalized",
            "materialized_dir": "${paths.data_dir}/" + name + "/materialized",
            "artifact_dir": "${paths.data_dir}/${dataset.dataset_family}/artifacts/${dataset.name}",
            "paths": {"embeddings": "${dataset.materialized_dir}/embeddings"},
            "num_topics": 2, "topic_pe": {"num_rounds": 2, "num_reverse_rounds": 2}})
    _dump(cfg / "data" / "retriever.yaml", {"_target_": "src.data.g_retrieval_datamodule.GRetrievalDataModule", "dataset_cfg": "${dataset}",
                                            "batch_size": 4, "drop_last": False,
                                            "splits": {"train": "train", "validation": "validation", "test": "test"}})
    _dump(cfg / "model" / "retriever_module.yaml", {
        "_target_": "src.models.retriever_module.RetrieverModule", "compile_model": True,
        "retriever": {"_target_": "src.models.components.retriever.Retriever", "emb_dim": emb_dim, "hidden_dim": hidden_dim,
                      "num_topics": "${dataset.num_topics}", "topic_pe": True, "direction_mode": "bidirectional",
                      "dde_cfg": {"num_rounds": "${dataset.topic_pe.num_rounds}", "num_reverse_rounds": "${dataset.topic_pe.num_reverse_rounds}"},
                      "dropout_p": 0.1},
        "loss": {"_target_": "src.losses.retriever_loss.RetrieverLoss", "path_weight": 0, "path_warmup_steps": 0,
                 "infonce_temperature": 0.07, "infonce_weight": 1.0, "bce_weight": 0.0},
        "evaluation_cfg": {"edge_recall_k": "${window.k_values}", "connectivity_k": "${window.k_values}", "bridge_metrics": True,
                           "feature_metrics": False, "ablate_topic": False, "emit_predict_outputs": False}})
    _dump(cfg / "callbacks" / "default.yaml", {"defaults": ["_self_"]})
    _dump(cfg / "callbacks" / "retriever_eval.yaml", {"defaults": ["g_agent_materializer", "retriever_topk_edge_writer", "_self_"]})
    _dump(cfg / "callbacks" / "retriever_topk_edge_writer.yaml", {"retriever_topk_edge_writer": {
        "_target_": "src.callbacks.retriever_topk_edge_writer.RetrieverTopKEdgeWriter",
        "output_dir": "${dataset.artifact_dir}/eval_retriever", "split": "${run.split}", "enabled": True,
        "artifact_name": "eval_retriever", "schema_version": 1, "topk_values": "${model.evaluation_cfg.edge_recall_k}",
        "textualize": False, "overwrite": True}})
    _dump(cfg / "callbacks" / "g_agent_materializer.yaml", {"g_agent_materializer": {
        "_target_": "src.callbacks.g_agent_materializer.GAgentMaterializationCallback",
        "settings": {"_target_": "src.data.components.g_agent_builder.GAgentSettings",
                     "enabled": "${oc.select:run.build_g_agent,true}", "edge_top_k": "${run.edge_top_k}",
                     "start_keep_ratio": "${oc.select:run.start_keep_ratio,0.25}", "max_hops": "${run.max_hops}",
                     "score_temperature": "${run.score_temperature}", "allow_empty_answer": "${oc.select:run.allow_empty_answer,false}",
                     "output_path": "${dataset.artifact_dir}/g_agent/${run.split}_g_agent.pt"},
        "lmdb_path": "${dataset.paths.embeddings}/${run.split}.lmdb"}})
    _dump(cfg / "trainer" / "predict.yaml", {"_target_": "lightning.pytorch.trainer.Trainer", "accelerator": "gpu", "devices": 1,
                                             "default_root_dir": "${paths.output_dir}"})
    _dump(cfg / "paths" / "default.yaml", {"root_dir": "${oc.env:EVI_TEST_PROJECT_ROOT,${oc.env:PWD}}", "data_dir": str(data_dir),
                                           "log_dir": "${paths.root_dir}/logs/", "output_dir": "${hydra:runtime.output_dir}",
                                           "work_dir": "${hydra:runtime.cwd}"})
    _dump(cfg / "hydra" / "default.yaml", {
        "defaults": [{"override hydra_logging": "colorlog"}],
        "run": {"dir": "${paths.log_dir}/${oc.select:hydra.runtime.choices.experiment,${task_name}}_"
                       "${oc.select:hydra.runtime.choices.dataset,${oc.select:dataset.name,unknown}}/runs/fixed"}})
    # the training side of the tree (configs/train.yaml, experiment/train_retriever.yaml, callbacks/{model_checkpoint,
    # early_stopping}.yaml, trainer/{default,gpu}.yaml of the reference, in miniature)
    _dump(cfg / "train.yaml", {
        "defaults": ["_self_", {"window": "default"}, {"ckpt": "default"}, {"dataset": None}, {"data": None}, {"model": None},
                     {"callbacks": "train_default"}, {"logger": None}, {"trainer": "default"}, {"paths": "default"}, {"hydra": "default"},
                     {"experiment": None}, {"optional local": "default"}],
        "task_name": "train", "tags": ["dev"], "train": True, "test": True, "ckpt_path": None, "seed": 42},
        header="# @package _global_\n\n")
    _dump(cfg / "callbacks" / "model_checkpoint.yaml", {"model_checkpoint": {
        "_target_": "lightning.pytorch.callbacks.ModelCheckpoint", "dirpath": None, "filename": None, "monitor": None, "save_last": None,
        "save_top_k": 1, "mode": "min", "auto_insert_metric_name": True, "save_weights_only": False}})
    _dump(cfg / "callbacks" / "early_stopping.yaml", {"early_stopping": {
        "_target_": "lightning.pytorch.callbacks.EarlyStopping", "monitor": None, "min_delta": 0.0, "patience": 3, "mode": "min"}})
    _dump(cfg / "callbacks" / "train_default.yaml", {
        "defaults": ["model_checkpoint", "early_stopping", "_self_"],
        "model_checkpoint": {"dirpath": "${paths.output_dir}/checkpoints", "filename": "epoch_{epoch:03d}", "monitor": "val/ranking/mrr",
                             "mode": "max", "save_last": True, "auto_insert_metric_name": False},
        "early_stopping": {"monitor": "val/ranking/mrr", "patience": 10, "mode": "max"}})
    _dump(cfg / "trainer" / "default.yaml", {"_target_": "lightning.pytorch.trainer.Trainer", "default_root_dir": "${paths.output_dir}",
                                             "min_epochs": 50, "max_epochs": 200, "accelerator": "auto", "devices": "auto",
                                             "precision": "16-mixed", "gradient_clip_val": 1.0, "check_val_every_n_epoch": 10})
    _dump(cfg / "trainer" / "gpu.yaml", {"defaults": ["default"], "accelerator": "gpu", "devices": 1})
    _dump(cfg / "experiment" / "train_retriever.yaml", {
        "defaults": [{"override /data": "retriever"}, {"override /model": "retriever_module"}, {"override /trainer": "gpu"},
                     {"override /logger": "none"}],
        "task_name": "retriever_train", "tags": ["retriever"],
        "trainer": {"min_epochs": 0, "max_epochs": 6, "check_val_every_n_epoch": 1},
        "model": {"compile_model": False, "loss": {"infonce_weight": 1.0, "bce_weight": 0.0},
                  "optimizer_cfg": {"type": "adamw", "lr": 3.0e-3, "weight_decay": 1.0e-4},
                  "scheduler_cfg": {"type": "cosine", "t_max": 6, "eta_min": 1.0e-6, "interval": "epoch"}},
        "callbacks": {"model_checkpoint": {"monitor": "val/answer/reachability@20", "mode": "max", "filename": "epoch_{epoch:03d}",
                                           "auto_insert_metric_name": False, "save_weights_only": True},
                      "early_stopping": {"monitor": "val/answer/reachability@20", "mode": "max", "patience": 10}}},
        header="# @package _global_\n#\n# Retriever training (single recommended config).\n\n")
    _dump(cfg / "logger" / "none.yaml", {})
    _dump(cfg / "experiment" / "eval_retriever.yaml", {
        "defaults": [{"override /data": "retriever"}, {"override /model": "retriever_module"}, {"override /callbacks": "retriever_eval"},
                     {"override /logger": "none"}, {"override /trainer": "predict"}],
        "model": {"compile_model": False, "evaluation_cfg": {"split": "${run.split}", "ablate_topic": False}},
        "run": {"name": "eval_retriever", "task_name": "eval/retriever", "tags": ["eval", "retriever"], "run_all_splits": True,
                "edge_top_k": 30, "max_hops": 3, "build_g_agent": False, "eval_mode": "test", "ckpt_path": "${ckpt.retriever}",
                "dataset_variants": ["${dataset.dataset_family}", "${dataset.dataset_family}-sub"], "require_dual_datasets": True},
        "ckpt": {"retriever": "${oc.env:EVI_TEST_RETRIEVER_CKPT,null}"},
        "data": {"splits": {"test": "${run.split}"}},
        "callbacks": {"retriever_topk_edge_writer": {"output_dir": "${dataset.artifact_dir}/eval_retriever", "textualize": False}}},
        header="# @package _global_\n#\n# overlay for the retriever evaluation run\n\n")
    return cfg
