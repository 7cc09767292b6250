"""Linear-operator wrappers used around the effective Hamiltonian (reference ``tenpy/linalg/sparse.py``):
``NpcLinearOperatorWrapper`` (:118), ``SumNpcLinearOperator`` (:152), ``ShiftNpcLinearOperator`` (:187),
``OrthogonalNpcLinearOperator`` (:220).  Every operation is a device call on block vectors (inner products, axpy)."""
from . import np_conserved as npc
from .krylov_based import gram_schmidt, iadd_prefactor_other

__all__ = ['NpcLinearOperatorWrapper', 'SumNpcLinearOperator', 'ShiftNpcLinearOperator', 'OrthogonalNpcLinearOperator']


class NpcLinearOperatorWrapper:
    """Base class: everything that is not overridden is looked up on the wrapped operator."""

    def __init__(self, orig_operator):
        self.orig_operator = orig_operator

    def __getattr__(self, name):
        if name == 'orig_operator':          # (not yet set: unpickling / copy)
            raise AttributeError(name)
        return getattr(self.orig_operator, name)

    def unwrapped(self):
        op = self.orig_operator
        return op.unwrapped() if isinstance(op, NpcLinearOperatorWrapper) else op

    def matvec(self, vec):
        raise NotImplementedError("subclasses implement matvec")


class SumNpcLinearOperator(NpcLinearOperatorWrapper):
    """``(A + B) |vec>``."""

    def __init__(self, orig_operator, other_operator):
        super().__init__(orig_operator)
        self.other_operator = other_operator

    def matvec(self, vec):
        res = self.orig_operator.matvec(vec)
        res.iadd_prefactor_other(1., self.other_operator.matvec(vec))
        return res


class ShiftNpcLinearOperator(NpcLinearOperatorWrapper):
    """``(H + shift) |vec>``."""

    def __init__(self, orig_operator, shift):
        super().__init__(orig_operator)
        self.shift = shift

    def matvec(self, vec):
        res = self.orig_operator.matvec(vec)
        res.iadd_prefactor_other(self.shift, vec)
        return res


class OrthogonalNpcLinearOperator(NpcLinearOperatorWrapper):
    """``H -> P H P`` with ``P = 1 - sum_o |o><o|`` for the (Gram-Schmidt ortho-normalised) ``ortho_vecs``."""

    def __init__(self, orig_operator, ortho_vecs):
        super().__init__(orig_operator)
        self.ortho_vecs = gram_schmidt(list(ortho_vecs))

    def matvec(self, vec):
        vec = vec.copy(deep=True)
        for o in self.ortho_vecs:
            iadd_prefactor_other(vec, -npc.inner(o, vec, axes='range', do_conj=True), o)
        vec = self.orig_operator.matvec(vec)
        for o in self.ortho_vecs[::-1]:
            iadd_prefactor_other(vec, -npc.inner(o, vec, axes='range', do_conj=True), o)
        return vec
