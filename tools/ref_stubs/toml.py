"""Import-time stand-in for toml (no TOML I/O is exercised when generating goldens)."""


def load(*a, **k):  # pragma: no cover
    raise RuntimeError("toml is not installed")


dump = load
TomlNumpyEncoder = object
