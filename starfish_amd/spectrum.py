"""
Data containers feeding the model (reference: Starfish/spectrum.py).  Only the data contract of the
hot path is kept -- masked ``wave / flux / sigma`` per order define the N pixels of an order; HDF5
I/O is available when ``h5py`` is installed (it is not on the GPU box) and plotting is out of scope.
"""

import numpy as np


class Order:
    """One echelle order with a boolean pixel mask (Starfish/spectrum.py:8-62)."""

    def __init__(self, _wave, _flux, _sigma=None, mask=None):
        self._wave = np.asarray(_wave)
        self._flux = np.asarray(_flux)
        self._sigma = np.zeros_like(self._flux) if _sigma is None else np.asarray(_sigma)
        self.mask = (
            np.ones_like(self._wave, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        )

    @property
    def wave(self):
        """numpy.ndarray : the masked wavelength array"""
        return self._wave[self.mask]

    @property
    def flux(self):
        """numpy.ndarray : the masked flux array"""
        return self._flux[self.mask]

    @property
    def sigma(self):
        """numpy.ndarray : the masked flux-uncertainty array"""
        return self._sigma[self.mask]

    def __len__(self):
        return len(self._wave)

    def __eq__(self, other):
        if not isinstance(other, Order):
            return NotImplemented
        return (
            np.array_equal(self._wave, other._wave)
            and np.array_equal(self._flux, other._flux)
            and np.array_equal(self._sigma, other._sigma)
            and np.array_equal(self.mask, other.mask)
        )

    def __repr__(self):
        return f"Order(npix={len(self)}, unmasked={int(self.mask.sum())})"


class Spectrum:
    """Rectangular multi-order spectrum (Starfish/spectrum.py:65-115).  1-D inputs become one order;
    ``sigmas`` default to ones and ``masks`` to all-True."""

    def __init__(self, waves, fluxes, sigmas=None, masks=None, name="Spectrum"):
        waves = np.atleast_2d(waves)
        fluxes = np.atleast_2d(fluxes)
        sigmas = np.atleast_2d(sigmas) if sigmas is not None else np.ones_like(fluxes)
        masks = (
            np.atleast_2d(masks).astype(bool) if masks is not None else np.ones_like(waves, dtype=bool)
        )
        assert fluxes.shape == waves.shape, "flux array incompatible shape."
        assert sigmas.shape == waves.shape, "sigma array incompatible shape."
        assert masks.shape == waves.shape, "mask array incompatible shape."
        self.orders = [Order(waves[i], fluxes[i], sigmas[i], masks[i]) for i in range(len(waves))]
        self.name = name

    def __getitem__(self, index):
        return self.orders[index]

    def __setitem__(self, index, order):
        if len(order) != len(self.orders[0]):
            raise ValueError("Invalid order length; no ragged spectra allowed")
        self.orders[index] = order

    def __len__(self):
        return len(self.orders)

    def __iter__(self):
        return iter(self.orders)

    # masked views
    @property
    def waves(self):
        return np.asarray([o.wave for o in self.orders])

    @property
    def fluxes(self):
        return np.asarray([o.flux for o in self.orders])

    @property
    def sigmas(self):
        return np.asarray([o.sigma for o in self.orders])

    # unmasked views
    @property
    def _waves(self):
        return np.asarray([o._wave for o in self.orders])

    @property
    def _fluxes(self):
        return np.asarray([o._flux for o in self.orders])

    @property
    def _sigmas(self):
        return np.asarray([o._sigma for o in self.orders])

    @property
    def masks(self):
        """The full 2-D boolean masks."""
        return np.asarray([o.mask for o in self.orders])

    @property
    def shape(self):
        """(norders, npixels); assigning reshapes following numpy rules (spectrum.py:185-199)."""
        return (len(self), len(self.orders[0]))

    @shape.setter
    def shape(self, shape):
        self.__dict__.update(self.reshape(shape).__dict__)

    def reshape(self, shape):
        """A reshaped copy (Starfish/spectrum.py:202-220)."""
        return self.__class__(
            self._waves.reshape(shape),
            self._fluxes.reshape(shape),
            self._sigmas.reshape(shape),
            self.masks.reshape(shape),
            name=self.name,
        )

    @classmethod
    def load(cls, filename):
        """Load from HDF5 with keys waves/fluxes/sigmas/masks (Starfish/spectrum.py:222-245); ``*.npz``: the numpy
        container `save` writes without h5py (same keys)."""
        if str(filename).endswith(".npz"):
            with np.load(filename, allow_pickle=False) as base:
                name = str(base["name"]) if "name" in base.files else None
                return cls(base["waves"], base["fluxes"], base["sigmas"], base["masks"], name=name)
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - h5py is absent on the GPU box
            raise ImportError("Spectrum.load needs h5py; build the Spectrum from arrays instead") from e
        with h5py.File(filename, "r") as base:
            name = base.attrs["name"] if "name" in base.attrs else None
            return cls(base["waves"][:], base["fluxes"][:], base["sigmas"][:], base["masks"][:], name=name)

    def save(self, filename):
        """Write to HDF5 (Starfish/spectrum.py:247-267), or to a numpy container when the name ends in ``.npz``."""
        if str(filename).endswith(".npz"):
            extra = {} if self.name is None else {"name": np.array(self.name)}
            np.savez_compressed(filename, waves=self._waves, fluxes=self._fluxes, sigmas=self._sigmas, masks=self.masks, **extra)
            return
        try:
            import h5py
        except ImportError as e:  # pragma: no cover
            raise ImportError("Spectrum.save needs h5py") from e
        with h5py.File(filename, "w") as base:
            base.create_dataset("waves", data=self._waves, compression=9)
            base.create_dataset("fluxes", data=self._fluxes, compression=9)
            base.create_dataset("sigmas", data=self._sigmas, compression=9)
            base.create_dataset("masks", data=self.masks, compression=9)
            if self.name is not None:
                base.attrs["name"] = self.name

    def __repr__(self):
        return f"{self.name} ({len(self)} orders)"
