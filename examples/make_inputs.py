#!/usr/bin/env python
"""Writes the input FILES the example drivers read, like the reference's drivers do (`water1024.pdb`, `mpidwater.xml` in the
working directory): a PDB (CRYST1 + HETATM records, waters in O, H1, H2 order) and an OpenMM-style MPID force-field XML.

    python examples/make_inputs.py          # -> examples/water_1024/{water1024.pdb, mpidwater.xml}
                                            #    examples/water_pol_1024/{water1024.pdb, mpidwater.xml}
                                            #    examples/openmm_api/{water1024.pdb, forcefield.xml}

Geometry of water_1024: the reference example's coordinates, held as data in tests/golden/p1_water1024.npz (three decimals,
so the PDB round trip is exact).  water_pol_1024: the seeded synthetic liquid box (the reference's shipped geometry has 0.67 A
contacts on which its own SCF diverges).  The XML carries the MPID water parameters of admp_amd/systems.py
(values of the reference's mpidwater.xml, in its units: nm, nm^2, nm^3), polarizabilities zero for the first example."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from admp_amd import systems as S          # noqa: E402


def write_pdb(path, positions, box):
    with open(path, 'w') as fh:
        fh.write('REMARK  water box for the ADMP example drivers\n')
        fh.write('CRYST1%9.3f%9.3f%9.3f%7.2f%7.2f%7.2f P 1           1\n' % (box[0, 0], box[1, 1], box[2, 2], 90, 90, 90))
        for i, (x, y, z) in enumerate(positions):
            name, el = (('O', 'O'), ('H1', 'H'), ('H2', 'H'))[i % 3]
            fh.write('HETATM%5d  %-3s HOH A%4d    %8.3f%8.3f%8.3f  1.00  0.00           %s  \n' %
                     ((i + 1) % 100000, name, (i // 3 + 1) % 10000, x, y, z, el))
        fh.write('END\n')


def write_xml(path, polarizable):
    o, h = S._O_CART, S._H_CART                  # c0, dX..dZ (e A), qXX qYY qZZ qXY qXZ qYZ (e A^2) as the drivers use them

    def mp(type_, kz, kx, c):                    # back to the file's units: nm (x10), nm^2 (x300, the drivers' factor)
        return ('   <Multipole type="%s" kz="%s" kx="%s" c0="%r" dX="%r" dY="%r" dZ="%r" qXX="%r" qYY="%r" qZZ="%r" '
                'qXY="%r" qXZ="%r" qYZ="%r"/>\n' % (type_, kz, kx, c[0], c[1] / 10, c[2] / 10, c[3] / 10, c[4] / 300,
                                                    c[5] / 300, c[6] / 300, c[7] / 300, c[8] / 300, c[9] / 300))
    pol = 0.00088 if polarizable else 0.0
    with open(path, 'w') as fh:
        fh.write('<ForceField>\n <AtomTypes>\n  <Type name="380" class="OW" element="O" mass="15.999"/>\n'
                 '  <Type name="381" class="HW" element="H" mass="1.008"/>\n </AtomTypes>\n'
                 ' <Residues>\n  <Residue name="HOH">\n   <Atom name="H1" type="381"/>\n   <Atom name="H2" type="381"/>\n'
                 '   <Atom name="O" type="380"/>\n   <Bond from="0" to="2"/>\n   <Bond from="1" to="2"/>\n  </Residue>\n'
                 ' </Residues>\n <MPIDForce>\n')
        fh.write(mp('380', '-381', '-381', o))
        fh.write(mp('381', '380', '381', h))
        fh.write('   <Polarize type="380" polarizabilityXX="%r" polarizabilityYY="%r" polarizabilityZZ="%r" thole="%r"/>\n'
                 % (pol, pol, pol, S.THOLE_O))
        fh.write('   <Polarize type="381" polarizabilityXX="0.0" polarizabilityYY="0.0" polarizabilityZZ="0.0" thole="0.0"/>\n')
        fh.write(' </MPIDForce>\n</ForceField>\n')


def write_forcefield_xml(path):
    """force field in the format of the reference's examples/openmm_api/forcefield.xml (<ADMPDispForce>, <ADMPPmeForce>):
    the per-type tables A, B, Q, C6, C8, C10 are the per-atom values of admp_amd/systems.py taken back through the unit
    conversions of admp/api.py:185-193; the multipoles / polarizabilities are the MPID water values in nm units."""
    o, h = S._O_CART, S._H_CART

    def disp(k):
        return ('A="%r" B="%r" Q="%r" C6="%r" C8="%r" C10="%r"' % (
            S.TT_A[k] * 2625.5, S.TT_B[k] / 0.0529177249, S.TT_Q[k], S.C6[k] ** 2 / 1e6, S.C8[k] ** 2 / 1e8, S.C10[k] ** 2 / 1e10))

    def mom(c):
        return ('c0="%r" dX="%r" dY="%r" dZ="%r" qXX="%r" qXY="%r" qYY="%r" qXZ="%r" qYZ="%r" qZZ="%r"' % (
            c[0], c[1] / 10, c[2] / 10, c[3] / 10, c[4] / 300, c[7] / 300, c[5] / 300, c[8] / 300, c[9] / 300, c[6] / 300))
    scales = ' '.join('%sScale1%d="%s"' % (k, i, '0.00' if i < 5 else '1.00') for k in ('m',) for i in range(2, 7))
    scales3 = ' '.join('%sScale1%d="%s"' % (k, i, '0.00' if i < 5 else '1.00') for k in ('m', 'p', 'd') for i in range(2, 7))
    with open(path, 'w') as fh:
        fh.write('<ForceField>\n <AtomTypes>\n  <Type name="380" class="OW" element="O" mass="15.999"/>\n'
                 '  <Type name="381" class="HW" element="H" mass="1.008"/>\n </AtomTypes>\n'
                 ' <Residues>\n  <Residue name="HOH">\n   <Atom name="H1" type="381"/>\n   <Atom name="H2" type="381"/>\n'
                 '   <Atom name="O" type="380"/>\n   <Bond from="0" to="2"/>\n   <Bond from="1" to="2"/>\n  </Residue>\n'
                 ' </Residues>\n')
        fh.write(' <ADMPDispForce %s>\n   <Atom type="380" %s/>\n   <Atom type="381" %s/>\n </ADMPDispForce>\n'
                 % (scales, disp(0), disp(1)))
        fh.write(' <ADMPPmeForce lmax="2" pmax="10" %s>\n' % scales3)
        fh.write('   <Atom type="380" kz="-381" kx="-381" %s/>\n   <Atom type="381" kz="380" kx="381" %s/>\n' % (mom(o), mom(h)))
        fh.write('   <Polarize type="380" polarizabilityXX="0.00088" polarizabilityYY="0.00088" polarizabilityZZ="0.00088" '
                 'thole="%r"/>\n' % S.THOLE_O)
        fh.write('   <Polarize type="381" polarizabilityXX="0.0" polarizabilityYY="0.0" polarizabilityZZ="0.0" thole="0.0"/>\n')
        fh.write(' </ADMPPmeForce>\n</ForceField>\n')


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'p1_water1024.npz'))
    d1 = os.path.join(ROOT, 'examples', 'water_1024')
    write_pdb(os.path.join(d1, 'water1024.pdb'), g['positions'], g['box'])
    write_xml(os.path.join(d1, 'mpidwater.xml'), polarizable=False)
    pos, box = S.synthetic_water_box(1024, seed=20240)
    d2 = os.path.join(ROOT, 'examples', 'water_pol_1024')
    write_pdb(os.path.join(d2, 'water1024.pdb'), pos, box)
    write_xml(os.path.join(d2, 'mpidwater.xml'), polarizable=True)
    d3 = os.path.join(ROOT, 'examples', 'openmm_api')
    os.makedirs(d3, exist_ok=True)
    write_pdb(os.path.join(d3, 'water1024.pdb'), pos, box)
    write_forcefield_xml(os.path.join(d3, 'forcefield.xml'))
    print('wrote', d1, d2, d3)


if __name__ == '__main__':
    main()
