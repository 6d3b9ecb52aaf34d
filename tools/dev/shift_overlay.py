#!/usr/bin/env python3
"""Development overlay of the shift-reuse loop generator (quakemigrate_amd/csrc/gen_shift_asm.py).

The product generator reads nothing from the environment.  For an A/B build of a
variant this script imports it, sets module constants named on the command line and prints the file:

    python tools/dev/shift_overlay.py PF_AHEAD=32 WIDE_SLOTTED=0 > build_variants/shift_asm_x.inc

Timing experiments (WRONG RESULTS by construction -- what a component of the loop costs in situ):
`DROP=<regex>` leaves out every emitted instruction that matches, `ONLY=<name regex>` restricts that to the
flavours whose function name matches, e.g. DROP='ds_read_b128' ONLY='shift_wide'.
`SUM_DEGREE=<6..10>`: the fused detect's 2^f polynomial at another degree (with -DQM_EXP2_DEGREE_SUM=<same>).

tools/shift_variants.sh builds a library per variant around such a file; nothing of it reaches
`__graft_entry__.build()`, which regenerates the committed qm_shift_asm.inc from the product constants
and ignores the environment."""
import importlib.util
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]


def main(argv):
    spec = importlib.util.spec_from_file_location("gen_shift_asm", ROOT / "quakemigrate_amd" / "csrc" / "gen_shift_asm.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    import re

    drop = only = None
    sum_degree = None
    for item in list(argv):
        name, _, value = item.partition("=")
        if name in ("DROP", "ONLY", "SUM_DEGREE"):
            argv.remove(item)
            if name == "DROP":
                drop = re.compile(value)
            elif name == "ONLY":
                only = re.compile(value)
            else:
                sum_degree = int(value)
    if sum_degree is not None:
        # (right results to that polynomial's accuracy: the 2^f of the fused detect's running sums -- degree 8 in
        # the product, 7.8e-13 -- at another degree; build the unit with -DQM_EXP2_DEGREE_SUM=<the same>)
        emit_product = gen.emit

        def emit_degree(degree, volume, *rest, **kw):
            emit_product(sum_degree if (degree == 8 and not volume) else degree, volume, *rest, **kw)
        gen.emit = emit_degree
    if drop is not None:
        emit = gen.emit
        plain = gen.Emitter.__call__

        def emit_filtered(degree, volume, lds_state, far, lazy, block, name, *rest, **kw):
            active = only is None or only.search(name)
            gen.Emitter.__call__ = (lambda self, text: None if drop.search(text) else plain(self, text)) if active else plain
            try:
                emit(degree, volume, lds_state, far, lazy, block, name, *rest, **kw)
            finally:
                gen.Emitter.__call__ = plain
        gen.emit = emit_filtered
    for item in argv:
        name, _, value = item.partition("=")
        if not hasattr(gen, name):
            raise SystemExit(f"shift_overlay: the generator has no constant {name}")
        old = getattr(gen, name)
        setattr(gen, name, type(old)(int(value)) if isinstance(old, (bool, int)) else type(old)(value))
    gen.WMAX = 4 * gen.NQMAX
    gen.OVERLAY = ",".join(sys.argv[1:]).replace('"', "'").replace("\\", "/") or "none"
    gen.main()


if __name__ == "__main__":
    main(sys.argv[1:])
