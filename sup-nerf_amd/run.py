"""``python -m supnerf_amd.run [--precision P] <script.py> [script arguments ...]`` -- run one of the reference's entry scripts
(``optimize_nuscenes.py``, ``optimize_kitti.py``, ``optimize_waymo.py``, ``train_nuscenes.py``, ``scripts/demo.py``) UNMODIFIED on the
MI355X path.

What it does, in this order: (1) ``supnerf_amd.install()`` -- the import hook that re-points ``utils.render_rays_v2`` ...,
``model_supnerf.SUPNeRF`` ..., ``renderer.NeRFRenderer`` ... as the script's own ``import`` statements load them (binding.py); (2) makes
``sys.argv`` and ``sys.path[0]`` what ``python <script.py> ...`` would have made them (the scripts compute ``src/`` from the working
directory themselves, optimize_nuscenes.py:1-3); (3) ``runpy.run_path(script, run_name='__main__')`` in THIS interpreter.  No GPU call is
made before the script starts and no process is replaced (no ``exec``), so the launcher is safe on pools that forbid re-exec after HIP
initialisation.  Run it from the reference's checkout (the scripts open ``jsonfiles/...`` relative to the working directory) with this
repository's root on ``PYTHONPATH``.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    precision = None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--precision" and len(argv) >= 2:
            precision, argv = argv[1], argv[2:]
        elif argv[0].startswith("--precision="):
            precision, argv = argv[0].split("=", 1)[1], argv[1:]
        elif argv[0] == "--":
            argv = argv[1:]
            break
        else:
            break
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = argv[0]
    if not os.path.isfile(script):
        print(f"supnerf_amd.run: no such script: {script}", file=sys.stderr)
        return 2
    from . import binding as I
    from . import model as M
    if precision is not None:
        if precision not in ("auto", "fp32", "bf16x3"):
            print(f"supnerf_amd.run: --precision must be auto, fp32 or bf16x3, not {precision!r}", file=sys.stderr)
            return 2
        M.DEFAULT_PRECISION = precision
    report = I.install()
    print(f"[supnerf_amd.run] installed (hook={report['hook']}, already-imported modules patched: {sorted(report['patched'])}); "
          f"running {script}", file=sys.stderr, flush=True)
    sys.argv = [script] + argv[1:]
    sys.path[0:0] = [os.path.dirname(os.path.abspath(script))]          # what ``python script.py`` puts first
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
