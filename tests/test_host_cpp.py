"""Builds and runs the C++ tests of the host-side mirror (kzero_amd/csrc/host): the reference's host side is compiled
Rust and there is no cargo in the image, so the mirror of the Network trait, job channel, executor loop, mappers and
symmetry wrapper is C++ and its tests are C++ programs."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(REPO, "tests", "cpp")
BUILD = os.path.join(CPP, "build")
GOLDEN = os.path.join(REPO, "tests", "golden")


def _build(src, out, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-pthread", os.path.join(CPP, src), "-o",
           os.path.join(BUILD, out), *extra]
    subprocess.check_call(cmd)
    return os.path.join(BUILD, out)


@pytest.mark.timeout(300)
def test_host_mirror_cpu_with_sanitizers():
    exe = _build("test_host.cpp", "test_host_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    out = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests ok" in out.stdout


@pytest.mark.timeout(300)
def test_host_mirror_cpu_thread_sanitizer():
    exe = _build("test_host.cpp", "test_host_tsan", ["-fsanitize=thread"])
    out = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout + out.stderr


def test_hip_network_test_compiles_against_the_c_abi():
    """The GPU test links only against libkzhip.so's C ABI (include/kz_hip.h): no torch, no HIP headers."""
    _build("test_hip_network.cpp", "test_hip_network",
           [f"-L{os.path.join(REPO, 'kzero_amd')}", "-lkzhip", f"-Wl,-rpath,{os.path.join(REPO, 'kzero_amd')}"])


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_hip_network_and_executor_loop_on_gpu():
    exe = _build("test_hip_network.cpp", "test_hip_network",
                 [f"-L{os.path.join(REPO, 'kzero_amd')}", "-lkzhip", f"-Wl,-rpath,{os.path.join(REPO, 'kzero_amd')}"])
    out = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "hip network tests ok" in out.stdout
