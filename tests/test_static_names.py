"""CPU suite: every name a function of bench.py / __graft_entry__.py / the package reads as a global is defined at module
level or is a builtin.  (Round 4: `bench.py --workload sdxl_refiner` died on a NameError in a code path no CPU test
executes; branches that need a GPU are still checked for names that cannot resolve.)"""
import builtins
import glob
import os
import symtable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ([os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
         + sorted(glob.glob(os.path.join(ROOT, "llm-groundeddiffusion_amd", "*.py")))
         + sorted(glob.glob(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin", "**", "*.py"), recursive=True)))
IMPLICIT = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__builtins__", "__class__"}


def unresolved(path):
    top = symtable.symtable(open(path).read(), path, "exec")
    module = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    star = "import *" in open(path).read()
    bad = []

    def walk(t):
        for ch in t.get_children():
            for s in ch.get_symbols():
                if s.is_global() and s.is_referenced() and not s.is_assigned():
                    n = s.get_name()
                    if n not in module and n not in IMPLICIT:
                        bad.append((ch.get_name(), ch.get_lineno(), n))
            walk(ch)
    walk(top)
    for s in top.get_symbols():                      # module-level reads of names nothing binds
        if s.is_referenced() and not (s.is_assigned() or s.is_imported() or s.is_namespace()) and s.get_name() not in IMPLICIT:
            bad.append(("<module>", 0, s.get_name()))
    return [] if star else bad


def test_no_unresolvable_global_names():
    assert len(FILES) > 20
    problems = {os.path.relpath(f, ROOT): unresolved(f) for f in FILES}
    problems = {f: b for f, b in problems.items() if b}
    assert not problems, problems
