"""Packaging and the command's entry (reference: setup.py:44-66 -- console script
``scvae=scvae.__main__:main`` -- and scvae/__main__.py:21-26), on the CPU."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pyproject():
    try:
        import tomllib as toml
    except ImportError:
        import tomli as toml
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as handle:
        return toml.load(handle)


def test_console_script_resolves_to_the_cli(capsys):
    project = _pyproject()["project"]
    target = project["scripts"]["scvae"]
    module_name, function_name = target.split(":")
    main = getattr(importlib.import_module(module_name), function_name)
    with pytest.raises(SystemExit) as exit_info:
        main(["--version"])
    assert exit_info.value.code == 0
    import scvae_amd
    assert capsys.readouterr().out.strip() == "scvae " + scvae_amd.__version__
    # the reference's two commands
    with pytest.raises(SystemExit):
        main(["train", "--help"])
    assert "--minibatch-size" in capsys.readouterr().out
    with pytest.raises(SystemExit):
        main(["evaluate", "--help"])
    assert "--model-versions" in capsys.readouterr().out


def test_package_metadata_and_build_step():
    config = _pyproject()
    assert config["tool"]["setuptools"]["dynamic"]["version"]["attr"] == (
        "scvae_amd.__version__")
    build = os.path.join(ROOT, config["tool"]["scvae_amd"]["build"])
    assert os.access(build, os.X_OK)
    # `python -m scvae_amd` is the same entry
    done = subprocess.run([sys.executable, "-m", "scvae_amd", "--version"],
                          cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and done.stdout.startswith("scvae ")


def _join_and_leave(rank, port, result):
    os.environ.update(WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SCVAE_DIST_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    from scvae_amd import cli
    assert cli._start_data_parallel() is True
    assert cli._start_data_parallel() is False     # already in a group: not ours to end
    value = torch.tensor([float(rank + 1)])
    dist.all_reduce(value)
    cli._stop_data_parallel()
    assert not dist.is_initialized()
    if rank == 0:
        with open(result, "w") as handle:
            handle.write(repr(value.item()))


def test_cli_joins_the_process_group_of_its_launcher(tmp_path):
    """The environment `python -m torch.distributed.run` leaves (WORLD_SIZE,
    RANK, LOCAL_RANK, MASTER_*) makes `cli.main` join and leave the group
    (gloo here; nccl = RCCL on the GPU box).  Without it: a single process."""
    from scvae_amd import cli
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(key, None)
    assert cli._start_data_parallel() is False
    import torch.multiprocessing as mp
    result = tmp_path / "sum.txt"
    port = 29300 + (os.getpid() % 300)
    mp.spawn(_join_and_leave, args=(port, str(result)), nprocs=2, join=True)
    assert float(result.read_text()) == 3.0
