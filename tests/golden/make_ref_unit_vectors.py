"""Extract the literal known-answer arrays of the reference's unit tests into JSON.

The reference's ``tests/test_sptial.py`` and ``tests/test_multipole.py`` hold the only
known-answer vectors that touch the PME path's helpers.  This script reads
those files AS TEXT (nothing is imported or executed from the reference), evaluates the
``pytest.mark.parametrize`` argument lists with ``array`` mapped to a plain list, and
writes ``ref_unit_vectors.json``.  Run in the build container only (needs /root/reference).
"""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


class _Arr:
    @staticmethod
    def array(x, *a, **k):
        return x


def extract(path):
    tree = ast.parse(open(path).read())
    out = {}
    for node in ast.walk(tree):
        if not isinstance(node, ast.FunctionDef):
            continue
        for dec in node.decorator_list:
            if isinstance(dec, ast.Call) and ast.unparse(dec.func).endswith('parametrize'):
                names = [s.strip() for s in ast.literal_eval(dec.args[0]).split(',')]
                cases = eval(compile(ast.Expression(dec.args[1]), path, 'eval'),
                             {'jnp': _Arr, 'np': _Arr})
                out[node.name] = [dict(zip(names, case)) for case in cases]
    return out


if __name__ == '__main__':
    data = {}
    data.update(extract(os.path.join(REF, 'tests', 'test_sptial.py')))
    data.update(extract(os.path.join(REF, 'tests', 'test_multipole.py')))
    with open(os.path.join(HERE, 'ref_unit_vectors.json'), 'w') as fh:
        json.dump(data, fh, indent=0)
    print({k: len(v) for k, v in data.items()})
