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


def test_bench_executor_chess_helpers():
    """The stand-ins bench_executor uses for the host work of a chess evaluation (pseudo-legal move generation, the
    SipHash-keyed move map of chess.rs:202-210): every generated move is one of the 1880 flat moves and both lookups agree."""
    exe = _build("test_bench_chess.cpp", "test_bench_chess_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "bench chess tests ok" in out.stdout, out.stdout + out.stderr


def test_bench_executor_compiles_against_the_c_abi():
    lib = os.path.join(REPO, "kzero_amd")
    _build("bench_executor.cpp", "bench_executor_check", [f"-L{lib}", "-lkzhip", f"-Wl,-rpath,{lib}"])


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


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_devices_in_one_process(tmp_path):
    """The Rust drop-in's topology (one process, a thread set per device: server.rs:323-331) on devices 0 and 1: both
    devices must give, bitwise, what device 0 gives alone — on the flagship one-launch f16 path (zero-copy pinned
    staging, two streams per engine, per-device weight cache), the multi-launch f32 path and the split-f16 path.
    Skipped on a box with one GPU; collected so that the first multi-GPU box runs it."""
    from kzero_amd import capi, synth
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs (this box has one): the test runs the day a multi-GPU node appears")
    lib = os.path.join(REPO, "kzero_amd")
    exe = _build("test_two_devices.cpp", "test_two_devices", [f"-L{lib}", "-lkzhip", f"-Wl,-rpath,{lib}"])
    big = tmp_path / "chess_2x256.kzm"
    big.write_bytes(synth.random_model("chess", 2, 256, "attention", seed=5))
    cases = [(str(big), "f16"), (str(big), "f32split16"), (os.path.join(GOLDEN, "go9_2x16_conv.kzm"), "f32"),
             (os.path.join(GOLDEN, "ataxx7_4x64.kzm"), "f16")]
    for model, dtype in cases:
        out = subprocess.run([exe, model, dtype], capture_output=True, text=True, timeout=200)
        assert out.returncode == 0, f"{model} {dtype}: " + out.stdout + out.stderr
        assert "two-device tests ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_thread_sets_in_one_process_rehearsal(tmp_path):
    """The same program on a ONE-GPU box with both thread sets on device 0 (KZ_TWO_DEVICES_REHEARSE=1): two job channels,
    2 x 2 pipelined executor threads with their own engines, one shared graph, both driven at once, a graph swap — bitwise
    what one thread set gives alone.  Everything of the two-device topology but the second piece of hardware."""
    from kzero_amd import synth
    lib = os.path.join(REPO, "kzero_amd")
    exe = _build("test_two_devices.cpp", "test_two_devices", [f"-L{lib}", "-lkzhip", f"-Wl,-rpath,{lib}"])
    big = tmp_path / "chess_2x256.kzm"
    big.write_bytes(synth.random_model("chess", 2, 256, "attention", seed=5))
    for model, dtype in [(str(big), "f16"), (str(big), "f32split16"), (os.path.join(GOLDEN, "go9_2x16_conv.kzm"), "f32"),
                         (os.path.join(GOLDEN, "ataxx7_4x64.kzm"), "f16")]:
        out = subprocess.run([exe, model, dtype], capture_output=True, text=True, timeout=200,
                             env={**os.environ, "KZ_TWO_DEVICES_REHEARSE": "1"})
        assert out.returncode == 0, f"{model} {dtype}: " + out.stdout + out.stderr
        assert "two-device tests ok" in out.stdout


def test_two_device_test_compiles_and_skips_without_two_gpus():
    """CPU side of the above: the program builds against the C ABI alone and reports 'skip' (exit code 77) where fewer
    than two GPUs are visible."""
    lib = os.path.join(REPO, "kzero_amd")
    exe = _build("test_two_devices.cpp", "test_two_devices", [f"-L{lib}", "-lkzhip", f"-Wl,-rpath,{lib}"])
    from kzero_amd import capi
    try:
        ndev = capi.device_count()
    except capi.KzError:
        ndev = 0
    if ndev >= 2:
        pytest.skip("two GPUs visible: test_two_devices_in_one_process runs the real thing")
    out = subprocess.run([exe, os.path.join(GOLDEN, "ataxx7_4x64.kzm")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 77 and "skip" in out.stdout, out.stdout + out.stderr
