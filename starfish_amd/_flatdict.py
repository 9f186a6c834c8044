"""
A small, dependency-free re-implementation of the subset of ``flatdict.FlatterDict``
that the Starfish ``SpectrumModel`` parameter store relies on
(reference usage: Starfish/models/spectrum_model.py:166,206,212,226,428-433,452,536,543).

Semantics kept:

* nested ``dict`` / ``list`` / ``tuple`` values are stored as child ``FlatterDict``
  objects; list items are keyed ``"0", "1", ...``;
* flat keys join the path with ``":"`` (``"local_cov:0:mu"``);
* ``__getitem__`` / ``__setitem__`` / ``__delitem__`` / ``__contains__`` accept both
  flat keys and parent keys; a parent key returns the (shared, mutable) child mapping;
* ``keys() / values() / items()`` are flat, in insertion order;
* ``as_dict()`` rebuilds the nested structure, restoring lists/tuples for children that
  were created from a sequence;
* assigning through a flat key whose parents do not exist creates them as mappings.
"""

from collections.abc import MutableMapping

_SEQ_TYPES = (list, tuple)


class FlatterDict(MutableMapping):
    """Insertion-ordered nested mapping addressed through delimiter-joined keys."""

    def __init__(self, value=None, delimiter=":"):
        self._delimiter = delimiter
        self._values = {}
        # the container type this node was built from (dict, list, tuple)
        self.original_type = dict
        if value is None:
            return
        if isinstance(value, FlatterDict):
            self.original_type = value.original_type
            for k, v in value._values.items():
                self._values[k] = self._wrap(v)
        elif isinstance(value, _SEQ_TYPES):
            self.original_type = type(value)
            for i, v in enumerate(value):
                self._values[str(i)] = self._wrap(v)
        elif isinstance(value, dict) or hasattr(value, "items"):
            for k, v in value.items():
                self[k] = v
        else:
            raise TypeError(f"cannot build FlatterDict from {type(value).__name__}")

    # ------------------------------------------------------------------ helpers
    def _wrap(self, v):
        if isinstance(v, FlatterDict):
            return FlatterDict(v, self._delimiter)
        if isinstance(v, (dict,) + _SEQ_TYPES):
            return FlatterDict(v, self._delimiter)
        return v

    def _split(self, key):
        key = str(key)
        if self._delimiter in key:
            head, rest = key.split(self._delimiter, 1)
            return head, rest
        return key, None

    # ------------------------------------------------------------- mapping core
    def __getitem__(self, key):
        head, rest = self._split(key)
        if head not in self._values:
            raise KeyError(key)
        node = self._values[head]
        if rest is None:
            return node
        if not isinstance(node, FlatterDict):
            raise KeyError(key)
        return node[rest]

    def __setitem__(self, key, value):
        head, rest = self._split(key)
        if rest is None:
            self._values[head] = self._wrap(value)
            return
        node = self._values.get(head)
        if not isinstance(node, FlatterDict):
            node = FlatterDict(delimiter=self._delimiter)
            self._values[head] = node
        node[rest] = value

    def __delitem__(self, key):
        head, rest = self._split(key)
        if head not in self._values:
            raise KeyError(key)
        if rest is None:
            del self._values[head]
            return
        node = self._values[head]
        if not isinstance(node, FlatterDict):
            raise KeyError(key)
        del node[rest]

    def __contains__(self, key):
        try:
            self[key]
        except (KeyError, TypeError):
            return False
        return True

    def keys(self):
        out = []
        for k, v in self._values.items():
            if isinstance(v, FlatterDict):
                sub = v.keys()
                if sub:
                    out.extend(self._delimiter.join((k, s)) for s in sub)
                else:
                    out.append(k)
            else:
                out.append(k)
        return out

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    # ---------------------------------------------------------------- exporting
    def as_dict(self):
        """Nested plain containers; children born from a list/tuple come back as one."""
        out = {}
        for k, v in self._values.items():
            if isinstance(v, FlatterDict):
                out[k] = v._seq_or_dict()
            else:
                out[k] = v
        return out

    def _seq_or_dict(self):
        if self.original_type in _SEQ_TYPES:
            return self.original_type(
                [
                    x._seq_or_dict() if isinstance(x, FlatterDict) else x
                    for x in self._values.values()
                ]
            )
        return self.as_dict()

    # --------------------------------------------------------------- comparison
    def __eq__(self, other):
        if isinstance(other, FlatterDict):
            return self.as_dict() == other.as_dict()
        if isinstance(other, dict):
            return self.as_dict() == other
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __repr__(self):
        return f"<FlatterDict {self.as_dict()!r}>"

    def __str__(self):
        return "{" + ", ".join(f"{k!r}: {v!r}" for k, v in self.items()) + "}"

    def copy(self):
        return FlatterDict(self, self._delimiter)
