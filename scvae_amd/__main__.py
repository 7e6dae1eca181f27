import sys

from scvae_amd.cli import main

if __name__ == "__main__":
    status = main()
    sys.exit(0 if not isinstance(status, int) else status)
