"""The compiled host side (wfa_amd/host/wfa.hpp, C++ mirror of the Go API) through the C-ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "cpp_host_test")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib = os.path.join(ROOT, "wfa_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", EXE, os.path.join(ROOT, "tests", "cpp_host_test.cpp"),
                           "-pthread", f"-L{lib}", "-lwfahip", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_host_builds_and_fails_loudly_without_gpu(built):
    """CPU: the header compiles against include/wfa_hip.h and links libwfahip.so; with no device the driver
    reports 'skipped' (exit 77) instead of computing anything on the CPU."""
    import torch
    _build()
    rc = subprocess.call([EXE])
    assert rc == (0 if torch.cuda.is_available() else 77)


@pytest.mark.gpu
def test_cpp_host_known_answers(built):
    _build()
    assert subprocess.call([EXE]) == 0
