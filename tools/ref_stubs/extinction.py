"""Import-time stand-in for the third-party `extinction` C extension (absent here).
Only lets `import Starfish.transforms` succeed; calling a law raises."""


def _absent(*a, **k):
    raise RuntimeError("`extinction` is not installed in this container (parity unpinned for Av != 0)")


ccm89 = odonnell94 = calzetti00 = fitzpatrick99 = fm07 = _absent
