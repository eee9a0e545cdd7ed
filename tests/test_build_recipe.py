"""The build recipe knows every file libellhip.so is made of: an edit to any of them must trigger a rebuild
(round 3 shipped two headers that `needs_build()` did not look at)."""
import os
import re

import ellalgo_rs_amd as pkg

INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure():
    """Every file reachable through `#include "..."` from the translation units, as absolute paths."""
    build = pkg.build
    todo = [os.path.join(build.CSRC, f) for f in build.SOURCES]
    seen = set()
    while todo:
        path = os.path.normpath(todo.pop())
        if path in seen:
            continue
        seen.add(path)
        with open(path) as fh:
            text = fh.read()
        for name in INC.findall(text):
            cand = os.path.normpath(os.path.join(os.path.dirname(path), name))
            if not os.path.exists(cand):
                cand = os.path.normpath(os.path.join(build.REPO_ROOT, "include", name))
            assert os.path.exists(cand), f"{path} includes {name}: not found"
            todo.append(cand)
    return seen


def test_every_included_file_is_a_build_dependency():
    build = pkg.build
    known = {os.path.normpath(os.path.join(build.CSRC, f)) for f in build.SOURCES + build.HEADERS}
    known |= {os.path.normpath(os.path.join(build.REPO_ROOT, "include", f)) for f in build.PUBLIC_HEADERS}
    missing = sorted(_closure() - known)
    assert not missing, f"build.py does not track {missing}"
    for must in ("group_kernels.hpp", "resident_kernels.hpp", "ell_kernels.hpp", "ellstable_kernels.hpp"):
        assert must in build.HEADERS


def test_an_edited_header_triggers_a_rebuild(tmp_path):
    build = pkg.build
    if not os.path.exists(build.LIB_PATH):
        return  # nothing built yet: needs_build() is trivially True
    t_lib = os.path.getmtime(build.LIB_PATH)
    for f in build.HEADERS:
        path = os.path.join(build.CSRC, f)
        st = os.stat(path)
        try:
            os.utime(path, (st.st_atime, t_lib + 10))
            assert build.needs_build(), f"touching {f} does not trigger a rebuild"
        finally:
            os.utime(path, (st.st_atime, st.st_mtime))
