"""potential_fn(positions, box, pairs, params) closures and their parameter gradients -- the calling convention of the
reference's front-end (admp/api.py:183-199 for the dispersion + Tang-Toennies potential, :442-455 for the multipolar
PME potential), without its OpenMM XML machinery: the calculators are built by the caller (ADMPPmeForce,
ADMPDispPmeForce, generate_pairwise_interaction) and wrapped here.

    pot_pme = pme_potential(pme_force, pol, tholes)
    E = pot_pme(positions, box, pairs, params)              # params: mScales, Q_local [, pScales, dScales, U_ind]
    g = param_gradient(pot_pme, positions, box, pairs, params)
    g['mScales'], g['Q_local'], g['pol'], g['tholes']       # jax.grad(pot_pme, argnums=3) of examples/openmm_api/run.py:44-46

`param_gradient` returns the entries that have hand-coded adjoints on this path: mScales, pScales,
dScales, Q_local, pol, tholes; there is no autodiff, so entries without one (the per-type A/B/Q/C tables of the
dispersion front-end) are absent rather than zero.
"""
import numpy as np


class _Potential:
    def __init__(self, energy, gradient):
        self._energy = energy
        self._gradient = gradient

    def __call__(self, positions, box, pairs, params):
        return self._energy(positions, box, pairs, params)


def pme_potential(pme_force, pol=None, tholes=None):
    """admp/api.py:442-455: params keys mScales, Q_local and, for a polarizable force, pScales, dScales, U_ind
    (pol / tholes are closed over, as in the reference)."""
    lpol = pme_force.lpol
    if lpol and (pol is None or tholes is None):
        raise ValueError('a polarizable force needs pol and tholes')

    def energy(positions, box, pairs, params):
        if lpol:
            return pme_force.get_energy(positions, box, pairs, params['Q_local'], pol, tholes, params['mScales'],
                                        params['pScales'], params['dScales'], U_init=params.get('U_ind'))
        return pme_force.get_energy(positions, box, pairs, params['Q_local'], params['mScales'])

    def gradient(positions, box, pairs, params):
        out = {'mScales': pme_force.get_mscale_gradient(positions, box, pairs, params['Q_local'], params['mScales'])}
        if lpol:
            _, _, dQ = pme_force.get_forces_and_dQ(positions, box, pairs, params['Q_local'], pol, tholes, params['mScales'],
                                                   params['pScales'], params['dScales'], U_init=params.get('U_ind'))
            out['pol'], out['tholes'] = pme_force.get_pol_thole_gradients(
                positions, box, pairs, params['Q_local'], pol, tholes, params['mScales'], params['pScales'],
                params['dScales'], U_init=pme_force.U_ind)
            out['pScales'] = pme_force.get_pscale_gradient(
                positions, box, pairs, params['Q_local'], pol, tholes, params['mScales'], params['pScales'],
                params['dScales'], U_init=pme_force.U_ind)
            out['dScales'] = np.zeros(len(out['pScales']))          # the reference ignores dScales (uscales = 1, pme.py:472)
        else:
            _, _, dQ = pme_force.get_forces_and_dQ(positions, box, pairs, params['Q_local'], params['mScales'])
        out['Q_local'] = dQ
        return out
    return _Potential(energy, gradient)


def disp_potential(disp_force, pair_interaction, map_atomtype):
    """admp/api.py:183-199: E = E_sr (Tang-Toennies) - E_lr (dispersion PME); params keys mScales and the per-atom-type
    tables A (kJ/mol), B (nm^-1), Q, C6, C8, C10 (kJ/mol nm^p) with the reference's unit conversions."""
    idx = np.asarray(map_atomtype)

    def lists(params):
        a = np.asarray(params['A'], dtype=np.float64)[idx] / 2625.5
        b = np.asarray(params['B'], dtype=np.float64)[idx] * 0.0529177249
        q = np.asarray(params['Q'], dtype=np.float64)[idx]
        c = np.stack([np.sqrt(np.asarray(params['C6'], dtype=np.float64)[idx] * 1e6),
                      np.sqrt(np.asarray(params['C8'], dtype=np.float64)[idx] * 1e8),
                      np.sqrt(np.asarray(params['C10'], dtype=np.float64)[idx] * 1e10)], axis=1)
        return a, b, q, c

    def energy(positions, box, pairs, params):
        a, b, q, c = lists(params)
        e_sr = pair_interaction(positions, box, pairs, params['mScales'], a, b, q, c[:, 0])
        e_lr = disp_force.get_energy(positions, box, pairs, c, params['mScales'])
        return e_sr - e_lr

    def gradient(positions, box, pairs, params):
        a, b, q, c = lists(params)
        g_sr = pair_interaction.get_mscale_gradient(positions, box, pairs, params['mScales'], a, b, q, c[:, 0])
        g_lr = disp_force.get_mscale_gradient(positions, box, pairs, c, params['mScales'])
        return {'mScales': g_sr - g_lr}
    return _Potential(energy, gradient)


def param_gradient(potential, positions, box, pairs, params):
    """dict of dE/dparams for the entries with an adjoint on this path -- the counterpart of
    jax.grad(potential, argnums=3)(positions, box, pairs, params) (examples/openmm_api/run.py:41-46)."""
    if not isinstance(potential, _Potential):
        raise TypeError('param_gradient takes a potential built by admp_amd.api')
    return potential._gradient(positions, box, pairs, params)
