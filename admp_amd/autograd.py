"""torch.autograd bridge: the PME energy as a differentiable node whose backward is the HIP adjoint.

The reference differentiates its energy with JAX not only w.r.t. positions but also w.r.t. parameters
(`grad(potential, argnums=3)` in examples/openmm_api/run.py:41-46), which is what force-field fitting and
geometry-dependent ("fluctuating") multipoles need.  Here the hot path returns dE/dpositions and dE/dQ_local
from hand-coded adjoints (admp_pme_energy_grad); this module exposes them to torch's autograd so that any
host-side model  Q_local = f(positions, theta)  is differentiated by the chain rule:

    E = pme_energy(pme_force, positions, box, pairs, Q_local, mScales)            # 0-d tensor
    E.backward()        # positions.grad = dE/dr + (dE/dQ)(dQ/dr),  theta.grad = (dE/dQ)(dQ/dtheta)

For a polarizable force the induced dipoles are converged first and held fixed in the backward pass
(Hellmann-Feynman, admp/pme.py:81-85), exactly like `value_and_grad(get_energy)` of the reference.
"""
import torch


class _PmeEnergy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, force, box, pairs, rest, U_init, positions, Q_local):
        with torch.no_grad():
            kw = {} if U_init is None else {'U_init': U_init}
            E, G, dQ = force.get_forces_and_dQ(positions, box, pairs, Q_local, *rest, **kw)
        G = torch.as_tensor(G, device=positions.device).to(positions.dtype)
        dQ = torch.as_tensor(dQ, device=Q_local.device).to(Q_local.dtype)
        ctx.save_for_backward(G, dQ)
        return torch.tensor(float(E), dtype=positions.dtype, device=positions.device)

    @staticmethod
    def backward(ctx, gout):
        G, dQ = ctx.saved_tensors
        return None, None, None, None, None, gout * G, gout * dQ


def pme_energy(force, positions, box, pairs, Q_local, *rest, U_init=None):
    """Differentiable electrostatic energy.  `rest` = (mScales,) or (pol, tholes, mScales, pScales, dScales),
    as in `ADMPPmeForce.get_energy`; positions and Q_local are torch tensors (either may require grad); U_init is the
    SCF start of a polarizable force (the reference's keyword of the same name)."""
    return _PmeEnergy.apply(force, box, pairs, tuple(rest), U_init, positions, Q_local)
