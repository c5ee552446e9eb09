"""Minimal stand-in for NamedTrajectories.jl's ``NamedTrajectory`` (external package of the
reference): the data layout the hot path reads: ``data`` is ``dim x N``, stored
column-major, ``datavec = vec(data)`` is knot-major and the NLP vector is ``[datavec; global_data]``
(src/solvers/evaluator.jl:230, 474-482)."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


class NamedTrajectory:
    def __init__(self, components, timestep, global_data=None, global_components=None):
        """components: ordered mapping name -> array (dim_i x N); timestep: name of the timestep
        COMPONENT (the integrators read ``traj.components[traj.timestep]``,
        bilinear_integrator.jl:123, so a fixed Float timestep is rejected)."""
        if not isinstance(timestep, str):
            raise ValueError("the engine requires a timestep component (Symbol), not a fixed Float")
        self.names = list(components.keys())
        arrays = [np.atleast_2d(np.asarray(components[k], dtype=np.float64)) for k in self.names]
        self.N = arrays[0].shape[1]
        for a in arrays:
            if a.shape[1] != self.N:
                raise ValueError("all components need the same number of knots")
        self.dims = OrderedDict((k, a.shape[0]) for k, a in zip(self.names, arrays))
        self.components = OrderedDict()
        off = 0
        for k, a in zip(self.names, arrays):
            self.components[k] = range(off, off + a.shape[0])
            off += a.shape[0]
        self.dim = off
        if timestep not in self.components or self.dims[timestep] != 1:
            raise ValueError("timestep must name a 1-dimensional component")
        self.timestep = timestep
        self.data = np.vstack(arrays)
        self.global_data = np.zeros(0) if global_data is None else np.asarray(global_data, dtype=np.float64).ravel()
        self.global_dim = self.global_data.size
        # NamedTrajectories' `global_components`: name -> indices into global_data (one unnamed block by default)
        if global_components is None:
            self.global_components = OrderedDict([("globals", range(self.global_dim))]) if self.global_dim else OrderedDict()
        else:
            self.global_components = OrderedDict((k, list(v)) for k, v in global_components.items())
            for v in self.global_components.values():
                if any(i < 0 or i >= self.global_dim for i in v):
                    raise ValueError("global component index out of range")

    @property
    def datavec(self):
        return np.ascontiguousarray(self.data.T).reshape(-1)

    def vec(self):
        """Z = [datavec; global_data]."""
        return np.concatenate([self.datavec, self.global_data])

    def update(self, Z):
        Z = np.asarray(Z, dtype=np.float64)
        self.data = Z[:self.dim * self.N].reshape(self.N, self.dim).T.copy()
        if self.global_dim:
            self.global_data = Z[self.dim * self.N:].copy()
