"""Composition of the reference's Hydra config tree without Hydra (neither `hydra` nor `omegaconf` is a dependency here).

The reference's entry point is `python src/eval.py experiment=eval_retriever dataset=webqsp ckpt.retriever=X`
(src/eval.py:1-40, configs/eval.yaml, configs/experiment/eval_retriever.yaml).  `evi_rag_amd.eval` accepts the same
command line against the SAME `configs/` directory; this module is the part of Hydra / OmegaConf that tree uses:

  * defaults lists — `_self_`, `group: option`, `group: null`, `optional group: option`, `override /group: option`,
    bare names relative to the file's own group; the `# @package _global_` header of the experiment overlays;
  * command-line overrides — `group=option` for a config group, `a.b.c=value` (YAML-typed) for a value,
    `+a.b=value` to add a key;
  * interpolation — `${a.b.c}`, `${oc.env:VAR,default}`, `${oc.select:key,default}`, `${hydra:runtime.output_dir}`,
    `${now:%Y-%m-%d}`, nested and embedded in strings, resolved lazily with cycle detection;
  * `instantiate` — `_target_` (+ `_partial_`) objects, recursively, with a table that maps the reference's class paths
    (`src.models.components.retriever.Retriever`, …) to the mirrors in this package.

Anything outside that subset raises ConfigError instead of being guessed at.
"""
from __future__ import annotations

import copy
import datetime
import functools
import importlib
import os
import re
from pathlib import Path
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple

import yaml


class ConfigError(ValueError):
    pass


# reference class path -> mirror in this package (same constructor kwargs)
REFERENCE_TARGETS: Dict[str, str] = {
    "src.models.components.retriever.Retriever": "evi_rag_amd.retriever.Retriever",
    "src.losses.retriever_loss.RetrieverLoss": "evi_rag_amd.loss.RetrieverLoss",
    "src.callbacks.retriever_topk_edge_writer.RetrieverTopKEdgeWriter": "evi_rag_amd.topk_writer.RetrieverTopKEdgeWriter",
    "src.data.components.g_agent_builder.GAgentSettings": "evi_rag_amd.g_agent.GAgentSettings",
    "src.data.components.g_agent_builder.GAgentBuilder": "evi_rag_amd.g_agent.GAgentBuilder",
    "src.metrics.retriever_metrics.EdgeRecallAtK": "evi_rag_amd.metrics.EdgeRecallAtK",
    "src.metrics.retriever_metrics.ScoreMargin": "evi_rag_amd.metrics.ScoreMargin",
    "src.metrics.reachability.AnswerReachability": "evi_rag_amd.metrics.AnswerReachability",
    "src.data.components.embedding_store.GlobalEmbeddingStore": "evi_rag_amd.embedding_store.GlobalEmbeddingStore",
}

_GLOBAL_PACKAGE = re.compile(r"^#\s*@package\s+_global_\s*$")


def _load(path: Path) -> Tuple[Dict[str, Any], bool]:
    text = path.read_text()
    is_global = False
    for line in text.splitlines():
        if not line.strip():
            continue
        if not line.lstrip().startswith("#"):
            break
        if _GLOBAL_PACKAGE.match(line.strip()):
            is_global = True
    body = yaml.safe_load(text)
    if body is None:
        body = {}
    if not isinstance(body, dict):
        raise ConfigError(f"{path}: top level must be a mapping")
    return body, is_global


def _merge(dst: Dict[str, Any], src: Mapping[str, Any]) -> Dict[str, Any]:
    """OmegaConf merge: mappings merge key by key, everything else (lists included) is replaced."""
    for k, v in src.items():
        if isinstance(v, Mapping) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _place(root: Dict[str, Any], package: Sequence[str], body: Mapping[str, Any]) -> None:
    node = root
    for part in package:
        nxt = node.get(part)
        if not isinstance(nxt, dict):
            nxt = node[part] = {}
        node = nxt
    _merge(node, body)


class _Entry:
    __slots__ = ("group", "option", "optional", "is_self")

    def __init__(self, group: Optional[str], option: Optional[str], optional: bool = False, is_self: bool = False) -> None:
        self.group, self.option, self.optional, self.is_self = group, option, optional, is_self


def _parse_defaults(raw: Any, where: str, own_group: Optional[str]) -> Tuple[List[_Entry], Dict[str, Optional[str]]]:
    """-> (entries in order, `override /group` requests found in this file)."""
    entries: List[_Entry] = []
    overrides: Dict[str, Optional[str]] = {}
    for item in raw or []:
        if isinstance(item, str):
            if item == "_self_":
                entries.append(_Entry(None, None, is_self=True))
            else:  # a sibling of this file in its own group
                if own_group is None:
                    raise ConfigError(f"{where}: bare defaults entry {item!r} outside a config group")
                entries.append(_Entry(own_group, item))
            continue
        if not isinstance(item, dict) or len(item) != 1:
            raise ConfigError(f"{where}: unsupported defaults entry {item!r}")
        (key, option), = item.items()
        key = str(key).strip()
        if "@" in key:
            raise ConfigError(f"{where}: package overrides in defaults ({key!r}) are not supported")
        if option is not None and not isinstance(option, str):
            raise ConfigError(f"{where}: defaults option for {key!r} must be a name or null, got {option!r}")
        if key.startswith("override "):
            overrides[key[len("override "):].strip().lstrip("/")] = option
            continue
        optional = key.startswith("optional ")
        if optional:
            key = key[len("optional "):].strip()
        group = key.lstrip("/") if key.startswith("/") or own_group is None else f"{own_group}/{key}"
        entries.append(_Entry(group, option, optional=optional))
    return entries, overrides


def _split_override(arg: str) -> Tuple[str, str, bool]:
    if "=" not in arg:
        raise ConfigError(f"override {arg!r} is not of the form key=value")
    key, value = arg.split("=", 1)
    add = key.startswith("+")
    key = key.lstrip("+")
    if key.startswith("~"):
        raise ConfigError(f"deleting keys ({arg!r}) is not supported")
    return key.strip(), value, add


def compose(config_dir: os.PathLike, config_name: str = "eval", overrides: Sequence[str] = (), *,
            hydra_runtime: Optional[Mapping[str, Any]] = None, searchpath: Sequence[os.PathLike] = ()) -> Dict[str, Any]:
    """The merged, fully interpolated config as plain dicts / lists."""
    raw, hydra_node = compose_raw(config_dir, config_name, overrides, hydra_runtime=hydra_runtime, searchpath=searchpath)
    return resolve_config(raw, hydra_node)


def resolve_config(raw: Mapping[str, Any], hydra_node: Dict[str, Any]) -> Dict[str, Any]:
    """Interpolate a composed tree (`raw` is left untouched, so a caller can edit keys — run.split, dataset — and
    resolve again, which is how the reference's entry point loops over splits and dataset variants)."""
    hydra_node = copy.deepcopy(hydra_node)
    scratch = dict(raw)
    scratch["hydra"] = hydra_node  # ${hydra.runtime.choices.x} / ${oc.select:hydra....} read it as a plain key
    if "output_dir" not in hydra_node["runtime"]:
        run_dir = select(hydra_node, "run.dir", None)
        hydra_node["runtime"]["output_dir"] = (str(_Resolver(scratch, hydra_node).value(run_dir)) if isinstance(run_dir, str)
                                               else os.getcwd())
    out = _Resolver(scratch, hydra_node).value(dict(raw))
    out.pop("hydra", None)
    return out


def _searchpath_dirs(value: Any) -> List[Path]:
    """`hydra.searchpath=[file:///abs/dir,...]` (Hydra's config search path): extra directories whose group files are found
    after the primary directory's.  Only `file://` entries (or plain paths) are supported."""
    items = value if isinstance(value, (list, tuple)) else [value]
    out = []
    for item in items:
        text = str(item)
        if text.startswith("pkg://"):
            raise ConfigError(f"hydra.searchpath entry {text!r}: pkg:// locations are not supported, use file://")
        out.append(Path(text[len("file://"):] if text.startswith("file://") else text))
    return out


def compose_raw(config_dir: os.PathLike, config_name: str = "eval", overrides: Sequence[str] = (), *,
                hydra_runtime: Optional[Mapping[str, Any]] = None,
                searchpath: Sequence[os.PathLike] = ()) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """(merged but NOT interpolated config, the hydra node that ${hydra:...} reads)."""
    root_dir = Path(config_dir)
    search_dirs: List[Path] = [root_dir] + [Path(d) for d in searchpath]
    rest = []
    for arg in overrides:
        if arg.lstrip("+").startswith("hydra.searchpath="):
            search_dirs += _searchpath_dirs(yaml.safe_load(arg.split("=", 1)[1]))
        else:
            rest.append(arg)
    overrides = rest

    def is_group(rel: str) -> bool:
        return any((d / rel).is_dir() for d in search_dirs)

    def find_option(group: str, option: str) -> Optional[Path]:
        for d in search_dirs:
            path = d / group / f"{option}.yaml"
            if path.exists():
                return path
        return None

    primary_path = root_dir / f"{config_name}.yaml"
    if not primary_path.exists():
        raise FileNotFoundError(f"primary config {primary_path} not found")
    primary, _ = _load(primary_path)
    entries, _ = _parse_defaults(primary.pop("defaults", None), str(primary_path), None)
    if not any(e.is_self for e in entries):
        entries.append(_Entry(None, None, is_self=True))

    group_choice: Dict[str, Optional[str]] = {}
    value_overrides: List[Tuple[str, Any, bool]] = []
    for arg in overrides:
        key, value, add = _split_override(arg)
        if is_group(key.replace(".", "/")) and "/" not in value:
            group_choice[key.replace(".", "/")] = None if value in ("null", "~", "") else value
        else:
            value_overrides.append((key, yaml.safe_load(value) if value != "" else "", add))

    listed = {e.group for e in entries if not e.is_self}
    for group in group_choice:
        if group not in listed:
            entries.append(_Entry(group, None))  # lenient: Hydra would insist on `+group=option`
            listed.add(group)

    # `override /group: option` lines sit in files selected by other choices (the experiment overlay): collect them first
    cache: Dict[Tuple[str, str], Tuple[Dict[str, Any], bool, List[_Entry], Dict[str, Optional[str]]]] = {}

    def load_option(group: str, option: str, optional: bool):
        key = (group, option)
        if key not in cache:
            path = find_option(group, option)
            if path is None:
                if optional:
                    cache[key] = ({}, False, [], {})
                    return cache[key]
                raise ConfigError(f"config group {group!r} has no option {option!r} "
                                  f"({', '.join(str(d / group / (option + '.yaml')) for d in search_dirs)} not found)")
            body, is_global = _load(path)
            sub_entries, sub_over = _parse_defaults(body.pop("defaults", None), str(path), group)
            if not any(e.is_self for e in sub_entries):
                sub_entries.append(_Entry(None, None, is_self=True))
            cache[key] = (body, is_global, sub_entries, sub_over)
        return cache[key]

    file_choice: Dict[str, Optional[str]] = {}

    def collect(group: str, option: str, optional: bool, depth: int = 0) -> bool:
        """`override /g: opt` requests of this option and of every option its own defaults list pulls in (an experiment
        that extends another experiment inherits that one's overrides, as in Hydra >= 1.1)."""
        if depth > 8:
            raise ConfigError(f"defaults nesting too deep at {group}/{option}")
        _, _, sub_entries, sub_over = load_option(group, option, optional)
        changed = False
        for g, opt in sub_over.items():
            if g not in listed:
                raise ConfigError(f"{group}/{option}: `override /{g}` but {g!r} is not in the primary defaults list")
            if file_choice.get(g, "\0") != opt:
                file_choice[g] = opt
                changed = True
        for se in sub_entries:
            if se.is_self or se.option is None:
                continue
            sub_option = group_choice.get(se.group, se.option) if se.group != group else se.option
            if sub_option is not None and se.group not in listed:
                changed |= collect(se.group, sub_option, se.optional, depth + 1)
            elif sub_option is not None and se.group == group:
                changed |= collect(se.group, sub_option, se.optional, depth + 1)
        return changed

    for _ in range(16):
        changed = False
        for e in entries:
            if e.is_self:
                continue
            option = group_choice.get(e.group, file_choice.get(e.group, e.option))
            if option is None or e.group == "hydra":
                continue
            changed |= collect(e.group, option, e.optional)
        if not changed:
            break
    else:
        raise ConfigError("defaults overrides do not settle (cyclic `override` entries?)")

    cfg: Dict[str, Any] = {}

    def merge_option(group: str, option: str, optional: bool, depth: int = 0) -> None:
        if depth > 8:
            raise ConfigError(f"defaults nesting too deep at {group}/{option}")
        body, is_global, sub_entries, _ = load_option(group, option, optional)
        package: List[str] = [] if is_global else group.split("/")  # Hydra's default package of a group file: its group
        for se in sub_entries:
            if se.is_self:
                _place(cfg, package, body)
            else:
                sub_option = group_choice.get(se.group, se.option) if se.group != group else se.option
                if sub_option is not None:
                    merge_option(se.group, sub_option, se.optional, depth + 1)

    # the `hydra` group configures Hydra itself (run directory, logging): it never becomes part of the job config.
    # Its run.dir is what ${hydra:runtime.output_dir} evaluates to.
    hydra_node: Dict[str, Any] = {"runtime": {"cwd": os.getcwd(), "choices": {}}, "job": {"name": config_name, "num": 0}}
    for e in entries:
        if e.is_self:
            _merge(cfg, primary)
            continue
        option = group_choice.get(e.group, file_choice.get(e.group, e.option))
        hydra_node["runtime"]["choices"][e.group] = option
        if option is None:
            continue
        if e.group == "hydra":
            path = find_option("hydra", option)
            if path is not None:
                body, _ = _load(path)
                body.pop("defaults", None)
                _merge(hydra_node, {k: v for k, v in body.items() if k not in ("runtime", "job")})
            continue
        merge_option(e.group, option, e.optional)

    for key, value, add in value_overrides:
        parts = key.split(".")
        node = cfg
        for part in parts[:-1]:
            nxt = node.get(part) if isinstance(node, dict) else None
            if not isinstance(nxt, dict):
                if nxt is None and (add or part not in node):
                    nxt = node[part] = {}
                else:
                    raise ConfigError(f"override {key!r}: {part!r} is not a mapping")
            node = nxt
        node[parts[-1]] = value

    if hydra_runtime:
        _merge(hydra_node, hydra_runtime)
    return cfg, hydra_node


# ---- interpolation ---------------------------------------------------------------------------------------------------
_MISSING = object()


def select(cfg: Any, key: str, default: Any = _MISSING) -> Any:
    node = cfg
    for part in key.split("."):
        if isinstance(node, Mapping) and part in node:
            node = node[part]
        elif isinstance(node, (list, tuple)) and part.lstrip("-").isdigit() and -len(node) <= int(part) < len(node):
            node = node[int(part)]
        else:
            if default is _MISSING:
                raise ConfigError(f"key {key!r} not found in the config")
            return default
    return node


def _split_top_level(text: str, sep: str = ",") -> List[str]:
    parts, depth, cur = [], 0, []
    i = 0
    while i < len(text):
        if text.startswith("${", i):
            depth += 1
            cur.append("${")
            i += 2
            continue
        ch = text[i]
        if ch == "}" and depth:
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
        i += 1
    parts.append("".join(cur))
    return parts


def _find_interpolations(text: str) -> List[Tuple[int, int]]:
    """(start, end) of every TOP-LEVEL ${...} in text (end exclusive)."""
    spans, i = [], 0
    while True:
        start = text.find("${", i)
        if start < 0:
            return spans
        depth, j = 0, start
        while j < len(text):
            if text.startswith("${", j):
                depth += 1
                j += 2
                continue
            if text[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        if depth != 0:
            raise ConfigError(f"unbalanced interpolation in {text!r}")
        spans.append((start, j + 1))
        i = j + 1


class _Resolver:
    def __init__(self, root: Any, hydra: Mapping[str, Any]) -> None:
        self.root, self.hydra, self.stack = root, hydra, []

    def value(self, node: Any) -> Any:
        if isinstance(node, str):
            return self.string(node)
        if isinstance(node, Mapping):
            return {k: self.value(v) for k, v in node.items()}
        if isinstance(node, (list, tuple)):
            return [self.value(v) for v in node]
        return node

    def string(self, text: str) -> Any:
        spans = _find_interpolations(text)
        if not spans:
            return text
        if len(spans) == 1 and spans[0] == (0, len(text)):
            return self.expression(text[2:-1])
        out, pos = [], 0
        for a, b in spans:
            out.append(text[pos:a])
            v = self.expression(text[a + 2: b - 1])
            out.append("null" if v is None else str(v))
            pos = b
        out.append(text[pos:])
        return "".join(out)

    def _literal(self, text: str) -> Any:
        text = text.strip()
        v = self.string(text) if "${" in text else text
        if not isinstance(v, str):
            return v
        if len(v) >= 2 and v[0] == v[-1] and v[0] in "\"'":
            return v[1:-1]
        try:
            return yaml.safe_load(v) if v != "" else ""
        except yaml.YAMLError:
            return v

    def expression(self, expr: str) -> Any:
        expr = expr.strip()
        head = expr.split(":", 1)[0]
        if ":" in expr and "${" not in head and re.fullmatch(r"[A-Za-z_][\w.]*", head):
            name, args_text = expr.split(":", 1)
            args = _split_top_level(args_text)
            if name == "oc.env":
                var = str(self._literal(args[0]))
                if var in os.environ:
                    return os.environ[var]
                if len(args) < 2:
                    raise ConfigError(f"environment variable {var!r} is not set and ${{oc.env:{var}}} has no default")
                return self._literal(",".join(args[1:]))
            if name == "oc.select":
                key = str(self._literal(args[0]))
                default = self._literal(",".join(args[1:])) if len(args) > 1 else None
                found = select(self.root, key, None)
                if found is None:
                    return default
                return self.lookup(key)
            if name == "hydra":
                return self.value(select(self.hydra, args_text.strip()))
            if name == "now":
                return datetime.datetime.now().strftime(args_text.strip())
            raise ConfigError(f"unsupported resolver {name!r} in ${{{expr}}}")
        key = self.string(expr) if "${" in expr else expr
        if not isinstance(key, str) or key.startswith("."):
            raise ConfigError(f"unsupported interpolation ${{{expr}}} (relative keys are not supported)")
        return self.lookup(key)

    def lookup(self, key: str) -> Any:
        if key in self.stack:
            raise ConfigError("interpolation cycle: " + " -> ".join(self.stack + [key]))
        self.stack.append(key)
        try:
            return self.value(select(self.root, key))
        finally:
            self.stack.pop()


def resolve_all(cfg: Any, *, hydra: Optional[Mapping[str, Any]] = None) -> Any:
    return _Resolver(cfg, hydra or {}).value(cfg)


# ---- instantiate -----------------------------------------------------------------------------------------------------
def locate(path: str, target_map: Optional[Mapping[str, str]] = None) -> Any:
    path = (REFERENCE_TARGETS if target_map is None else target_map).get(path, path)
    module, _, name = path.rpartition(".")
    if not module:
        raise ConfigError(f"_target_ {path!r} is not a dotted path")
    try:
        return getattr(importlib.import_module(module), name)
    except (ImportError, AttributeError) as exc:
        raise ConfigError(f"cannot locate _target_ {path!r}: {exc}") from exc


def instantiate(node: Any, *, target_map: Optional[Mapping[str, str]] = None, **extra: Any) -> Any:
    """hydra.utils.instantiate for the subset the reference's configs use: nested `_target_` mappings are built
    depth-first, `_partial_: true` returns functools.partial, `extra` kwargs override the node's."""
    if isinstance(node, (list, tuple)):
        return [instantiate(v, target_map=target_map) for v in node]
    if not isinstance(node, Mapping):
        return node
    if "_target_" not in node:
        return {k: instantiate(v, target_map=target_map) for k, v in node.items()}
    fn = locate(str(node["_target_"]), target_map)
    kwargs = {k: instantiate(v, target_map=target_map) for k, v in node.items()
              if k not in ("_target_", "_partial_", "_recursive_", "_convert_")}
    kwargs.update(extra)
    if node.get("_partial_"):
        return functools.partial(fn, **kwargs)
    return fn(**kwargs)


__all__ = ["ConfigError", "REFERENCE_TARGETS", "compose", "compose_raw", "resolve_config", "resolve_all", "select",
           "instantiate", "locate"]
