"""Minimal TOML writer (+ tomli-based reader) for SpectrumModel.save / load
(reference format: Starfish/models/spectrum_model.py:592-633; the `toml` package is not available)."""
import datetime as _dt

import numpy as np


def _scalar(v):
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if v != v:
            return "nan"
        if v in (float("inf"), float("-inf")):
            return "inf" if v > 0 else "-inf"
        r = repr(v)
        return r if any(c in r for c in ".en") else r + ".0"
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (_dt.datetime, _dt.date, _dt.time)):
        return v.isoformat()  # RFC 3339, a native TOML value (metadata commonly carries a date)
    raise TypeError(f"cannot write {type(v).__name__} to TOML")


def _is_table_list(v):
    return isinstance(v, (list, tuple)) and len(v) > 0 and all(isinstance(x, dict) for x in v)


def _emit(table, prefix, lines):
    subtables, table_lists = [], []
    for k, v in table.items():
        if isinstance(v, dict):
            subtables.append((k, v))
        elif _is_table_list(v):
            table_lists.append((k, v))
        elif isinstance(v, (list, tuple, np.ndarray)):
            lines.append(f"{k} = [" + ", ".join(_scalar(x) for x in v) + "]")
        else:
            lines.append(f"{k} = {_scalar(v)}")
    for k, v in subtables:
        name = f"{prefix}{k}"
        lines.append("")
        lines.append(f"[{name}]")
        _emit(v, name + ".", lines)
    for k, items in table_lists:
        name = f"{prefix}{k}"
        for item in items:
            lines.append("")
            lines.append(f"[[{name}]]")
            _emit(item, name + ".", lines)


def _quote_keys(d):
    """TOML bare keys may contain only A-Za-z0-9_-; the numeric cheb keys ("1") are fine."""
    return d


def dumps(doc):
    lines = []
    _emit(_quote_keys(doc), "", lines)
    return "\n".join(lines).lstrip("\n") + "\n"


def load(filename):
    try:
        import tomllib as _t  # py311+
    except ImportError:
        import tomli as _t
    with open(filename, "rb") as fh:
        return _t.load(fh)
