"""A deterministic stand-in for `openmm.app.Simulation`.  TEST INFRASTRUCTURE ONLY.

`sample_with_model` (utils/evaluation_utils.py:439-466, 558-565, 594-602, 623-626) drives the Simulation it is handed
through five calls - `context.setPositions`, `context.setVelocities`, `step`, `context.getState(getPositions=True,
getVelocities=True)` and `state.getPositions(asNumpy=True)._value` / `getVelocities` - and nothing else, so the MD
engine behind them does not matter for the parity of the loop.  This one integrates a damped pull towards the centroid in
float64 (OpenMM hands float64 arrays back as well); `oracle/gen_golden.py` passes it to the REAL reference function to
record tests/golden/mh_tiny_openmm.npz, and the tests pass it to the oracle and to the product.
"""
import numpy as np


class _Quantity:
    def __init__(self, value):
        self._value = value


class _State:
    def __init__(self, pos, vel):
        self._pos, self._vel = pos, vel

    def getPositions(self, asNumpy=False):
        return _Quantity(self._pos.copy())

    def getVelocities(self, asNumpy=False):
        return _Quantity(self._vel.copy())


class _Context:
    def __init__(self):
        self.pos = self.vel = None
        self.allow_thermal = False

    def setPositions(self, p):
        self.pos = np.array(p, dtype=np.float64)
        assert self.pos.ndim == 2 and self.pos.shape[1] == 3, self.pos.shape

    def setVelocities(self, v):
        self.vel = np.array(v, dtype=np.float64)
        assert self.vel.shape == self.pos.shape

    def setVelocitiesToTemperature(self, temperature):
        # sample_on_single_conditional (evaluation_utils.py:376-378) asks for thermal velocities when it treats the
        # conditioning velocities as resampled; the MH loop never does.  A fixed, temperature-scaled pattern.
        if not self.allow_thermal:
            raise AssertionError("the MH loop always passes velocities")
        i = np.arange(self.pos.size, dtype=np.float64).reshape(self.pos.shape)
        self.vel = 0.05 * np.sqrt(float(temperature) / 300.0) * np.cos(0.7 * i + 0.3)

    def getState(self, getPositions=False, getVelocities=False, **kwargs):
        return _State(self.pos, self.vel)


class _Integrator:
    def getTemperature(self):
        return 310.0


class FakeSimulation:
    def __init__(self, dt=0.004, pull=6.0, damping=0.95, allow_thermal=False):
        self.context = _Context()
        self.context.allow_thermal = allow_thermal
        self.integrator = _Integrator()
        self.dt, self.pull, self.damping = dt, pull, damping
        self.calls = 0

    def step(self, n):
        c = self.context
        for _ in range(int(n)):
            c.vel = self.damping * c.vel - self.dt * self.pull * (c.pos - c.pos.mean(axis=0, keepdims=True))
            c.pos = c.pos + self.dt * c.vel
        self.calls += 1
