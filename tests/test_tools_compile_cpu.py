"""Every command-line tool and script of the repository at least compiles (they run on the GPU box only, where a syntax error costs a call)."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "monodetr_amd", "tools", "*.py")) + glob.glob(os.path.join(ROOT, "scripts", "**", "*.py"), recursive=True)
               + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, ROOT) for f in FILES])
def test_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)
