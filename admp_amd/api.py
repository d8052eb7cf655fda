"""Force-field front-end -- counterpart of the reference's admp/api.py: an XML force field is read into GENERATORS that hold the
parameters (`generator.params`), and `Hamiltonian.createPotential` turns them into potentials in the calling convention

    E = potential_fn(positions, box, pairs, params)                       # admp/api.py:183-199 (dispersion), :442-455 (PME)
    g = param_gradient(potential_fn, positions, box, pairs, params)       # jax.grad(potential_fn, argnums=3)

    H = Hamiltonian('forcefield.xml')                                     # examples/openmm_api/run.py:17-46
    disp_generator, pme_generator = H.getGenerators()
    pot_disp, pot_pme = H.createPotential(Topology.from_pdb('water1024.pdb'), nonbondedCutoff=4.0)
    E = pot_disp(positions, box, pairs, disp_generator.params)
    g = param_gradient(pot_disp, positions, box, pairs, disp_generator.params)     # mScales, A, B, Q, C6, C8, C10

The reference builds this on OpenMM's ForceField / Topology classes; here the XML (`<Residues>`, `<ADMPDispForce>`,
`<ADMPPmeForce>` elements, same attributes) and the PDB are read by this package (ElementTree, admp_amd.parser.read_pdb) --
no OpenMM.  The calculators behind the potentials are the HIP ones (ADMPPmeForce, ADMPDispPmeForce,
generate_pairwise_interaction); there is no autodiff, every entry of `param_gradient` comes from a hand-coded adjoint:
mScales / pScales / dScales (class sums), Q_local (dE/dQ_local), pol / tholes (Thole sums), and for the dispersion potential
the per-type tables A, B, Q, C6, C8, C10 (per-atom sums of admp_disp_param_grad / admp_tt_param_grad chained through the unit
conversions of admp/api.py:185-193).  `pme_potential` / `disp_potential` wrap calculators the caller has built himself.
"""
import collections
import xml.etree.ElementTree as ET

import numpy as np

from . import parser as _parser
from .systems import convert_cart2harm


class _Potential:
    def __init__(self, energy, gradient):
        self._energy = energy
        self._gradient = gradient

    def __call__(self, positions, box, pairs, params):
        return self._energy(positions, box, pairs, params)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def pme_potential(pme_force, pol=None, tholes=None):
    """admp/api.py:442-455: params keys mScales, Q_local and, for a polarizable force, pScales, dScales, U_ind
    (pol / tholes are closed over, as in the reference; params['pol'] / params['tholes'] take precedence when present)."""
    lpol = pme_force.lpol

    def pt(params):
        return (params['pol'] if 'pol' in params else pol), (params['tholes'] if 'tholes' in params else tholes)

    def energy(positions, box, pairs, params):
        if lpol:
            p, t = pt(params)
            return pme_force.get_energy(positions, box, pairs, params['Q_local'], p, t, params['mScales'],
                                        params['pScales'], params['dScales'], U_init=params.get('U_ind'))
        return pme_force.get_energy(positions, box, pairs, params['Q_local'], params['mScales'])

    def gradient(positions, box, pairs, params):
        out = {'mScales': pme_force.get_mscale_gradient(positions, box, pairs, params['Q_local'], params['mScales'])}
        if lpol:
            p, t = pt(params)
            _, _, dQ = pme_force.get_forces_and_dQ(positions, box, pairs, params['Q_local'], p, t, params['mScales'],
                                                   params['pScales'], params['dScales'], U_init=params.get('U_ind'))
            out['pol'], out['tholes'] = pme_force.get_pol_thole_gradients(
                positions, box, pairs, params['Q_local'], p, t, params['mScales'], params['pScales'],
                params['dScales'], U_init=pme_force.U_ind)
            out['pScales'] = pme_force.get_pscale_gradient(
                positions, box, pairs, params['Q_local'], p, t, params['mScales'], params['pScales'],
                params['dScales'], U_init=pme_force.U_ind)
            out['dScales'] = np.zeros(len(out['pScales']))          # the reference ignores dScales (uscales = 1, pme.py:472)
            out['U_ind'] = np.zeros((pme_force.n_atoms, 3))         # stop_gradient on the SCF start (admp/pme.py:81-85)
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
        a = np.asarray(_np(params['A']), dtype=np.float64)[idx] / 2625.5
        b = np.asarray(_np(params['B']), dtype=np.float64)[idx] * 0.0529177249
        q = np.asarray(_np(params['Q']), dtype=np.float64)[idx]
        c = np.stack([np.sqrt(np.asarray(_np(params['C6']), dtype=np.float64)[idx] * 1e6),
                      np.sqrt(np.asarray(_np(params['C8']), dtype=np.float64)[idx] * 1e8),
                      np.sqrt(np.asarray(_np(params['C10']), dtype=np.float64)[idx] * 1e10)], axis=1)
        return a, b, q, c

    def energy(positions, box, pairs, params):
        a, b, q, c = lists(params)
        e_sr = pair_interaction(positions, box, pairs, params['mScales'], a, b, q, c[:, 0].copy())
        e_lr = disp_force.get_energy(positions, box, pairs, c, params['mScales'])
        return e_sr - e_lr

    def gradient(positions, box, pairs, params):
        a, b, q, c = lists(params)
        c6 = c[:, 0].copy()
        g_sr = pair_interaction.get_mscale_gradient(positions, box, pairs, params['mScales'], a, b, q, c6)
        g_lr = disp_force.get_mscale_gradient(positions, box, pairs, c, params['mScales'])
        out = {'mScales': g_sr - g_lr}
        # per-atom adjoints, chained to the per-type tables through admp/api.py:185-193
        da, db, dq, dc6 = (np.asarray(_np(x), dtype=np.float64) for x in
                           pair_interaction.get_param_gradient(positions, box, pairs, params['mScales'], a, b, q, c6))
        dc = np.zeros((len(idx), 3))
        lr = np.asarray(_np(disp_force.get_param_gradient(positions, box, pairs, c, params['mScales'])), dtype=np.float64)
        dc[:, :lr.shape[1]] = -lr
        dc[:, 0] += dc6
        ntype = len(np.asarray(_np(params['A'])))

        def per_type(v):
            return np.bincount(idx, weights=v, minlength=ntype)
        with np.errstate(divide='ignore', invalid='ignore'):
            half = [np.where(c[:, k] != 0.0, 0.5 / c[:, k], 0.0) for k in range(3)]     # d sqrt(x) / dx = 1 / (2 sqrt(x))
        out['A'] = per_type(da) / 2625.5
        out['B'] = per_type(db) * 0.0529177249
        out['Q'] = per_type(dq)
        out['C6'] = per_type(dc[:, 0] * half[0]) * 1e6
        out['C8'] = per_type(dc[:, 1] * half[1]) * 1e8
        out['C10'] = per_type(dc[:, 2] * half[2]) * 1e10
        return out
    return _Potential(energy, gradient)


def param_gradient(potential, positions, box, pairs, params):
    """dict of dE/dparams -- the counterpart of jax.grad(potential, argnums=3)(positions, box, pairs, params)
    (examples/openmm_api/run.py:41-46)."""
    if not isinstance(potential, _Potential):
        raise TypeError('param_gradient takes a potential built by admp_amd.api')
    return potential._gradient(positions, box, pairs, params)


# ------------------------------------------------------------------------------------------------ topology
class Topology:
    """What the generators need of a molecular system: atom names, residues, the periodic cell -- the role of
    openmm.app.PDBFile(...).topology in the reference's driver (examples/openmm_api/run.py:19)."""

    def __init__(self, names, res_names, res_seqs, box=None, positions=None):
        self.names, self.res_names, self.res_seqs = list(names), list(res_names), list(res_seqs)
        self.n_atoms = len(self.names)
        self.box = None if box is None else np.asarray(box, dtype=np.float64).reshape(3, 3)      # Angstrom, rows
        self.positions = None if positions is None else np.asarray(positions, dtype=np.float64)  # Angstrom
        self.residues = collections.OrderedDict()                     # (resSeq, running block) -> atom indices
        block, prev = 0, None
        for i, key in enumerate(zip(self.res_seqs, self.res_names)):
            if key != prev:
                block += 1
                prev = key
            self.residues.setdefault((block,) + key, []).append(i)

    @classmethod
    def from_pdb(cls, path):
        info = _parser.read_pdb(path)
        box = None
        if info.get('box'):
            a, b, c, al, be, ga = info['box']
            al, be, ga = np.radians([al, be, ga])
            bx = b * np.cos(ga)
            by = b * np.sin(ga)
            cx = c * np.cos(be)
            cy = c * (np.cos(al) - np.cos(be) * np.cos(ga)) / np.sin(ga)
            cz = np.sqrt(max(c * c - cx * cx - cy * cy, 0.0))
            box = np.array([[a, 0.0, 0.0], [bx, by, 0.0], [cx, cy, cz]])
            box[np.abs(box) < 1e-12] = 0.0
        return cls(info['names'], info['resNames'], info['resSeqs'], box, info['positions'])


def build_covalent_map(n_atoms, bonds, max_neighbor=6):
    """covalent_map[i, j] = number of bonds between i and j (1 .. max_neighbor), 0 beyond (admp/api.py:24-43), as a scipy CSR
    matrix (`.toarray()` gives the reference's dense form).  Breadth-first search from every atom."""
    import scipy.sparse as sp
    nbrs = [[] for _ in range(n_atoms)]
    for i, j in bonds:
        nbrs[i].append(j)
        nbrs[j].append(i)
    rows, cols, vals = [], [], []
    for i in range(n_atoms):
        if not nbrs[i]:
            continue
        dist = {i: 0}
        frontier = [i]
        for d in range(1, max_neighbor + 1):
            nxt = []
            for a in frontier:
                for b in nbrs[a]:
                    if b not in dist:
                        dist[b] = d
                        nxt.append(b)
            frontier = nxt
        for j, d in dist.items():
            if d > 0:
                rows.append(i)
                cols.append(j)
                vals.append(d)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n_atoms, n_atoms), dtype=np.int32)


def _cutoff_angstrom(x):
    """a plain number is Angstrom; an OpenMM-style quantity (`_value`, `unit`) in angstrom or nanometer is converted"""
    if hasattr(x, '_value') and hasattr(x, 'unit'):
        u = str(x.unit).lower()
        if u in ('nanometer', 'nm'):
            return 10.0 * float(x._value)
        if u in ('angstrom', 'a'):
            return float(x._value)
        raise ValueError('nonbondedCutoff: unit %s not understood (pass Angstrom as a float)' % u)
    return float(x)


class _Typed:
    """atom types of a topology from the <Residues> templates of the force field (atom name -> type, bonds)"""

    def __init__(self, templates, topology):
        self.types = [None] * topology.n_atoms
        self.bonds = []
        self.residue_of = [None] * topology.n_atoms
        for key, atoms in topology.residues.items():
            tmpl = templates.get(key[2])
            if tmpl is None:
                raise KeyError('no <Residue name="%s"> template in the force field' % key[2])
            by_name = {topology.names[i]: i for i in atoms}
            for name, typ in tmpl['atoms']:
                if name not in by_name:
                    raise KeyError('residue %s %s lacks atom %s of its template' % (key[2], key[1], name))
                self.types[by_name[name]] = typ
            for a, b in tmpl['bonds']:
                self.bonds.append((by_name[tmpl['atoms'][a][0]], by_name[tmpl['atoms'][b][0]]))
            for i in atoms:
                self.residue_of[i] = atoms
        missing = [i for i, t in enumerate(self.types) if t is None]
        if missing:
            raise KeyError('atoms without a template entry: %s ...' % missing[:5])


# ------------------------------------------------------------------------------------------------ generators
class ADMPDispGenerator:
    """<ADMPDispForce>: Tang-Toennies short-range term minus dispersion PME (admp/api.py:120-209).  params: mScales and the
    per-type tables A, B, Q, C6, C8, C10 in the XML's units."""

    def __init__(self, hamiltonian):
        self.ff = hamiltonian
        self.params = {}
        self.types = []
        self.ethresh = 1.0e-5
        self.pmax = 10
        self._potential = None

    @staticmethod
    def parseElement(element, hamiltonian):
        g = ADMPDispGenerator(hamiltonian)
        hamiltonian.registerGenerator(g)
        tab = collections.defaultdict(list)
        for atom in element.findall('Atom'):
            g.types.append(atom.attrib['type'])
            for k in ('A', 'B', 'Q', 'C6', 'C8', 'C10'):
                tab[k].append(float(atom.attrib[k]))
        g.params = {k: np.array(v) for k, v in tab.items()}
        g.params['mScales'] = np.array([float(element.attrib['mScale1%d' % i]) for i in range(2, 7)])
        g.types = np.array(g.types)

    def createForce(self, topology, typed, rc):
        from .disp_pme import ADMPDispPmeForce
        from .pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel
        self.map_atomtype = np.array([int(np.where(self.types == t)[0][0]) for t in typed.types])
        self.covalent_map = build_covalent_map(topology.n_atoms, typed.bonds, 6)
        self.disp_force = ADMPDispPmeForce(topology.box, self.covalent_map, rc, self.ethresh, self.pmax)
        self.pair_force = generate_pairwise_interaction(TT_damping_qq_c6_kernel, self.covalent_map, static_args={})
        self._potential = disp_potential(self.disp_force, self.pair_force, self.map_atomtype)

    def getJaxPotential(self):          # the reference's name for it
        return self._potential

    getPotential = getJaxPotential


class ADMPPmeGenerator:
    """<ADMPPmeForce>: multipolar (optionally polarizable) PME (admp/api.py:216-463).  params after createPotential: mScales,
    pScales, dScales, Q_local (Na,9), pol, tholes, U_ind -- per-ATOM arrays in the calculator's units, as in the reference."""

    _MOMENTS = ('c0', 'dX', 'dY', 'dZ', 'qXX', 'qYY', 'qZZ', 'qXY', 'qXZ', 'qYZ')

    def __init__(self, hamiltonian):
        self.ff = hamiltonian
        self.params = {}
        self.types = []
        self.kStrings = {'kz': [], 'kx': [], 'ky': []}
        self._input = collections.defaultdict(list)
        self.ethresh = 1.0e-5
        self.lmax = 2
        self.lpol = False
        self.ref_dip = ''           # optional file of induced dipoles (nm), one atom per line: the SCF start U_ind
        self._potential = None

    @staticmethod
    def parseElement(element, hamiltonian):
        g = ADMPPmeGenerator(hamiltonian)
        g.lmax = int(element.attrib.get('lmax', 2))
        g.pmax = int(element.attrib.get('pmax', 10))
        hamiltonian.registerGenerator(g)
        for key in ('mScale', 'pScale', 'dScale'):
            g.params[key + 's'] = np.array([float(element.attrib['%s1%d' % (key, i)]) for i in range(2, 7)])
        polar = {p.attrib['type']: p.attrib for p in element.findall('Polarize')}
        g.lpol = bool(polar)
        for atom in element.findall('Atom'):
            a = dict(atom.attrib)
            a.update({k: v for k, v in polar.get(a['type'], {}).items() if k != 'type'})
            g.types.append(a.pop('type'))
            for k in ('kz', 'kx', 'ky'):
                g.kStrings[k].append(a.pop(k, ''))
            for k, v in a.items():
                g._input[k].append(float(v))
        g._input = {k: np.array(v) for k, v in g._input.items()}
        g.types = np.array(g.types)

    def createForce(self, topology, typed, rc):
        from .pme import ADMPPmeForce
        n = topology.n_atoms
        mt = self.map_atomtype = np.array([int(np.where(self.types == t)[0][0]) for t in typed.types])
        p = self._input
        scale = (1.0, 10.0, 10.0, 10.0, 300.0, 300.0, 300.0, 300.0, 300.0, 300.0)      # e, e nm -> e A, e nm^2 -> 3 x e A^2
        Q = np.zeros((n, 10))
        for k, (name, s) in enumerate(zip(self._MOMENTS, scale)):
            if name in p:
                Q[:, k] = p[name][mt] * s                                            # admp/api.py:317-327
        self.params['Q_local'] = convert_cart2harm(Q, 2)
        if self.lpol:
            pol = np.stack([p['polarizabilityXX'][mt], p['polarizabilityYY'][mt], p['polarizabilityZZ'][mt]], axis=1)
            self.params['pol'] = (np.float32(1000) * pol.astype(np.float32).mean(axis=1)).astype(np.float64)   # nm^3 -> A^3, in
                                                                     # float32 like the reference (admp/api.py:330-332)
            self.params['tholes'] = p['thole'][mt].astype(np.float32).astype(np.float64)
            U = np.zeros((n, 3))
            if self.ref_dip:
                U = 10.0 * np.loadtxt(self.ref_dip)[:n, :3]
            self.params['U_ind'] = U
        # local frames: anchor TYPES from kz / kx / ky, resolved to atoms of the same residue (admp/api.py:391-411)
        axis_types, axis_indices = [], []
        for i in range(n):
            t = {'type': self.types[mt[i]], 'kz': self.kStrings['kz'][mt[i]], 'kx': self.kStrings['kx'][mt[i]],
                 'ky': self.kStrings['ky'][mt[i]]}
            _parser._axis_rule(t)
            slots = [s if s != '' else -1 for s in t['axis_indices'][1:]]
            for j in typed.residue_of[i]:
                if j == i:
                    continue
                for k, s in enumerate(slots):
                    if s == typed.types[j]:
                        slots[k] = j
                        break
            axis_types.append(t['axisType'])
            axis_indices.append([s if isinstance(s, (int, np.integer)) else -1 for s in slots])
        self.axis_types, self.axis_indices = np.array(axis_types), np.array(axis_indices)
        self.covalent_map = build_covalent_map(n, typed.bonds, 6)
        self.pme_force = ADMPPmeForce(topology.box, self.axis_types, self.axis_indices, self.covalent_map, rc, self.ethresh,
                                      self.lmax, self.lpol)
        self._potential = pme_potential(self.pme_force)

    def getJaxPotential(self):
        return self._potential

    getPotential = getJaxPotential


class Hamiltonian:
    """The force field of an XML file as a list of generators; createPotential(topology) returns their potentials in the
    order of the XML (admp/api.py:469-488)."""

    parsers = {'ADMPDispForce': ADMPDispGenerator.parseElement, 'ADMPPmeForce': ADMPPmeGenerator.parseElement}

    def __init__(self, xmlname):
        self._forces = []
        self._potentials = []
        root = ET.parse(xmlname).getroot()
        self._templates = {}
        for r in root.iter('Residue'):
            atoms = [(a.get('name'), a.get('type')) for a in r.findall('Atom')]
            if not atoms:
                continue                               # (a bond-definition file lists residues without typed atoms)
            names = [a[0] for a in atoms]
            bonds = []
            for b in r.findall('Bond'):      # template indices (OpenMM force-field files) or atom names
                ends = [b.get('from', b.get('atomName1')), b.get('to', b.get('atomName2'))]
                bonds.append(tuple(int(e) if e.lstrip('-').isdigit() else names.index(e) for e in ends))
            self._templates[r.get('name')] = {'atoms': atoms, 'bonds': bonds}
        for child in root:
            if child.tag in self.parsers:
                self.parsers[child.tag](child, self)

    def registerGenerator(self, generator):
        self._forces.append(generator)

    def getGenerators(self):
        return list(self._forces)

    def createPotential(self, topology, nonbondedMethod=None, nonbondedCutoff=10.0):
        """nonbondedCutoff in Angstrom (the reference passes an OpenMM quantity, rc * unit.angstrom)."""
        if topology.box is None:
            raise ValueError('the topology has no periodic cell (CRYST1 record)')
        rc = _cutoff_angstrom(nonbondedCutoff)
        typed = _Typed(self._templates, topology)
        self._potentials = []
        for g in self._forces:
            g.createForce(topology, typed, rc)
            self._potentials.append(g.getJaxPotential())
        return list(self._potentials)
