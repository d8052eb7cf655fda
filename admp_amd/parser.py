"""Readers for the two input files of the reference's example drivers -- a PDB with a CRYST1 record and an OpenMM-style
MPID force-field XML -- returning the structures those drivers consume (reference admp/parser.py: read_pdb :80-175,
read_xml :247-328, init_residues :377-460, assemble_covalent :462-476).  Host-side set-up only; nothing here is on the
timed path.  Written for this package (ElementTree, one pass, breadth-first covalent distances); the reference's own
parser is not shipped.

    pdbinfo = read_pdb('water1024.pdb')                       # serials (0-based), names, resNames, resSeqs, positions, box, charges
    atomTemplate, residueTemplate = read_xml('mpidwater.xml')
    atomDicts, residueDicts = init_residues(serials, names, resNames, resSeqs, positions, charges, atomTemplate, residueTemplate)
    covalent_map = assemble_covalent(residueDicts, n_atoms)   # (Na, Na) ints, 0 = not bonded within 4 bonds
"""
import collections
import xml.etree.ElementTree as ET

import numpy as np

__all__ = ['read_pdb', 'read_xml', 'init_residues', 'assemble_covalent', 'Atom', 'Residue']

# axis types (reference admp/spatial.py:44-74 numbering)
ZThenX, Bisector, ZBisect, ThreeFold, Zonly, NoAxisType = range(6)


def read_pdb(path):
    """ATOM / HETATM records up to END and the CRYST1 cell: dict of parallel lists (reference keys)."""
    out = collections.defaultdict(list)
    cell = None
    with open(path) as fh:
        for line in fh:
            rec = line[:6]
            if rec == 'CRYST1':
                cell = [float(line[6:15]), float(line[15:24]), float(line[24:33]), float(line[33:40]), float(line[40:47]),
                        float(line[47:54])]
            elif rec in ('ATOM  ', 'HETATM'):
                out['serials'].append(len(out['serials']))            # 0-based running index, like the reference
                out['names'].append(line[12:16].strip())
                out['resNames'].append(line[17:21].strip())
                out['resSeqs'].append(int(line[22:26]))
                out['positions'].append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
                out['elements'].append(line[76:78].strip().upper())
                ch = line[78:80].strip()
                out['charges'].append(ch or 0)
            elif rec.startswith('END'):
                break
    res = dict(out)
    res['positions'] = np.array(res['positions'], dtype=np.float64)
    res['box'] = cell                                                  # a, b, c, alpha, beta, gamma
    return res


def _axis_rule(tmpl):
    """axis type + the atom TYPES of the z / x / y anchors from the kz / kx / ky attributes (a leading '-' marks the
    bisector-style rules; reference set_axis_type admp/parser.py:177-245)"""
    keys = [tmpl.get(k, '') for k in ('kz', 'kx', 'ky')]
    neg = [k.startswith('-') for k in keys]
    typ = [k.lstrip('-') for k in keys]
    kz, kx, ky = typ
    axis = ZThenX
    if not kz:
        axis = NoAxisType
    if kz and not kx:
        axis = Zonly
    if (kz and neg[0]) or (kx and neg[1]):
        axis = Bisector
    if kx and neg[1] and ky and neg[2]:
        axis = ZBisect
    if kz and neg[0] and kx and neg[1] and ky and neg[2]:
        axis = ThreeFold
    tmpl['axisType'] = axis
    tmpl['axis_indices'] = [tmpl['type']] + typ                        # resolved to atom serials by init_residues


def read_xml(path):
    """(atom templates, residue templates): per residue-template atom its name, type, MPID multipoles (c0, dX.., qXX..),
    polarizabilities / thole, kz / kx / ky anchors and the derived axisType; per residue its atoms and bond list."""
    root = ET.parse(path).getroot()
    atoms, residues = [], []
    for r in root.iter('Residue'):
        rt = {'resName': r.get('name'), 'atoms': [], 'topo': collections.defaultdict(list)}
        for a in r.findall('Atom'):
            t = {'name': a.get('name'), 'type': a.get('type')}
            rt['atoms'].append(t)
            atoms.append(t)
        for b in r.findall('Bond'):
            rt['topo'][b.get('from')].append(b.get('to'))
        rt['topo'] = dict(rt['topo'])
        residues.append(rt)
    for m in root.iter('Multipole'):
        vals = {}
        for k, v in m.attrib.items():
            if k in ('kz', 'kx', 'ky'):
                vals[k] = v
            elif k != 'type':
                vals[k] = float(v)
        for k in ('kz', 'kx', 'ky'):
            vals.setdefault(k, '')
        for t in atoms:
            if t['type'] == m.get('type'):
                t.update(vals)
    for p in root.iter('Polarize'):
        vals = {k: p.get(k) for k in ('polarizabilityXX', 'polarizabilityYY', 'polarizabilityZZ', 'thole')}
        for t in atoms:
            if t['type'] == p.get('type'):
                t.update(vals)
    for t in atoms:
        _axis_rule(t)
    return atoms, residues


class Atom:
    def __init__(self, serial, name, resName, resSeq, position, charge):
        self.serial, self.name, self.resName, self.resSeq = serial, name, resName, resSeq
        self.position, self.charge = position, charge
        self.linkAtom = []

    def link(self, other):
        if other not in self.linkAtom:
            self.linkAtom.append(other)
        if self not in other.linkAtom:
            other.linkAtom.append(self)

    def __repr__(self):
        return '< Atom%s: %s >' % (self.serial, self.name)


class Residue:
    def __init__(self, resName, resSeq):
        self.resName, self.resSeq = resName, resSeq
        self.atoms = {}
        self.covalent_map = {}

    def add(self, serial, atom):
        self.atoms[serial] = atom

    def __getitem__(self, name):
        return next((a for a in self.atoms.values() if a.name == name), None)

    def __repr__(self):
        return '< Residue%s: %s >' % (self.resSeq, self.resName)


def init_residues(serials, names, resNames, resSeqs, positions, charges, atomTemplates, residueTemplates):
    """Atoms (template attributes copied on), grouped into residues; bonds from the residue template; anchor atom TYPES
    resolved to the serials of atoms of the same residue (each other atom claims the first anchor slot of its type, in
    residue order); covalent distances 1..4 inside the residue by breadth-first search."""
    by_name = {}
    for t in atomTemplates:
        by_name.setdefault(t['name'], t)
    res_tmpl = {rt['resName']: rt for rt in residueTemplates}
    atomDicts, residueDicts = {}, {}
    for serial, name, resName, resSeq, position, charge in zip(serials, names, resNames, resSeqs, positions, charges):
        atom = Atom(serial, name, resName, resSeq, position, charge)
        for k, v in by_name.get(name, {}).items():
            setattr(atom, k, list(v) if isinstance(v, list) else v)
        atomDicts[serial] = atom
        residueDicts.setdefault(resSeq, Residue(resName, resSeq)).add(serial, atom)
    for res in residueDicts.values():
        tmpl = res_tmpl[res.resName]
        for c, partners in tmpl['topo'].items():
            ca = res[tmpl['atoms'][int(c)]['name']]
            for p in partners:
                ca.link(res[tmpl['atoms'][int(p)]['name']])
        for atom in res.atoms.values():
            slots = [t if t != '' else -1 for t in atom.axis_indices[1:]]
            for other in res.atoms.values():
                if other.serial == atom.serial:
                    continue
                for i, s in enumerate(slots):
                    if s == other.type:
                        slots[i] = other.serial
                        break
            atom.axis_indices = slots
        for atom in res.atoms.values():                                # bond distances 1..4 by BFS over linkAtom
            dist = {atom.serial: 0}
            frontier = [atom]
            for d in (1, 2, 3, 4):
                nxt = []
                for a in frontier:
                    for b in a.linkAtom:
                        if b.serial not in dist:
                            dist[b.serial] = d
                            nxt.append(b)
                frontier = nxt
            res.covalent_map[atom.serial] = {s: d for s, d in dist.items() if d > 0}
    return atomDicts, residueDicts


def assemble_covalent(residueDicts, natoms):
    """dense (Na, Na) covalent_map; beyond ~50k atoms pass a scipy.sparse matrix to the calculators instead."""
    cov = np.zeros((natoms, natoms), dtype=int)
    for res in residueDicts.values():
        for i, row in res.covalent_map.items():
            for j, d in row.items():
                cov[i][j] = d
    return cov
