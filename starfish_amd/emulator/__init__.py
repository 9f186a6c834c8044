from .emulator import Emulator  # noqa: F401

__all__ = ["Emulator"]
