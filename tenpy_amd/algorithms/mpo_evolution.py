"""Time evolution with an MPO approximation of ``exp(-i dt H)`` -- reference ``tenpy/algorithms/mpo_evolution.py``
(``ExpMPOEvolution`` :30): the W_II operators of Zaletel et al. are applied to the MPS (one block tensordot per site) and
the state is compressed by a QR sweep and a truncating SVD sweep.  ``order`` 1: one MPO with step dt; ``order`` 2: two MPOs
with the complex steps (1 +- i)/2 dt."""
from ..linalg.truncation import TruncationError

__all__ = ['ExpMPOEvolution']


class ExpMPOEvolution:
    """Options: ``dt`` (0.1), ``N_steps`` (1), ``order`` (2), ``approximation`` ('II'), ``compression_method`` ('SVD'),
    ``trunc_params``."""

    def __init__(self, psi, model_H, options):
        self.psi = psi
        self.H = model_H
        self.options = dict(options)
        self.evolved_time = 0.
        self.trunc_err = TruncationError()
        self._U_MPO = None
        self._U_param = {}

    def calc_U(self, dt, order=2, approximation='II'):
        param = dict(dt=dt, order=order, approximation=approximation)
        if self._U_param == param:
            return
        self._U_param = param
        if order == 1:
            self._U_MPO = [self.H.make_U(dt * -1j, approximation)]
        elif order == 2:
            self._U_MPO = [self.H.make_U(-(1. + 1j) / 2. * dt * 1j, approximation),
                           self.H.make_U(-(1. - 1j) / 2. * dt * 1j, approximation)]
        else:
            raise ValueError("order %r not implemented" % (order,))

    def evolve_step(self, dt):
        trunc_err = TruncationError()
        for U in self._U_MPO:
            trunc_err = trunc_err + U.apply(self.psi, self.options)
        return trunc_err

    def evolve(self, N_steps, dt):
        self.calc_U(dt, self.options.get('order', 2), self.options.get('approximation', 'II'))
        trunc_err = TruncationError()
        for _ in range(N_steps):
            trunc_err = trunc_err + self.evolve_step(dt)
        self.evolved_time = self.evolved_time + N_steps * dt
        self.trunc_err = self.trunc_err + trunc_err
        return trunc_err

    def run(self):
        """``N_steps`` steps of ``dt``; ``preserve_norm`` (default True, real time evolution): put the norm of the state back
        afterwards (reference ``TimeEvolutionAlgorithm.run_evolution``)."""
        old_norm = self.psi.norm
        self.evolve(self.options.get('N_steps', 1), self.options.get('dt', 0.1))
        if self.options.get('preserve_norm', True):
            self.psi.norm = old_norm
        return self.psi
