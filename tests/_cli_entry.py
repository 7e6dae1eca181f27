"""Entry used by the data-parallel CLI tests: `scvae` (scvae_amd.cli.main) with the
shuffle seeded -- the reference's shuffle is unseeded (va:985), the tests compare
runs.  python -m torch.distributed.run ... tests/_cli_entry.py train <arguments>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import numpy
    numpy.random.seed(int(os.environ.get("SCVAE_TEST_SHUFFLE_SEED", "11")))
    from scvae_amd.cli import main
    status = main(sys.argv[1:])
    sys.exit(0 if not isinstance(status, int) else status)
