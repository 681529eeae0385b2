"""A plain C99 program (examples/c_driver.c) written against include/libxsmm.h the way users of the reference write
their drivers: it must compile with `gcc -std=c99 -pedantic`, link against libxsmm_amd.so, and -- on a GPU box --
reproduce its own gold loops through the unmodified dispatch -> param -> call flow (BASELINE config #1:
samples/xgemm f32 23x23x23; packed CSR 35x35)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libxsmm_amd", "lib")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cdrv") / "c_driver")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_driver.c"),
           "-L" + LIBDIR, "-lxsmm_amd", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _run(exe, *args):
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)


def test_c_header_compiles_as_c99_and_fails_loudly_without_a_gpu(driver):
    import torch
    r = _run(driver, "probe")
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        assert "devices=0 handle=NULL" in r.stdout and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("gemm", "23", "23", "23", "1"), ("gemm", "32", "32", "32", "8"), ("gemm", "64", "48", "80", "3"), ("spmm", "64"), ("spmm", "1000")])
def test_c_driver_reproduces_its_gold_loops_on_the_gpu(driver, args):
    r = _run(driver, *args)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "normf_rel" in r.stdout
