"""CPU: wave-level emulation of the MFMA tile index math (tests/host/test_tile_index.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_index_emulation(tmp_path):
    exe = str(tmp_path / "test_tile_index")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "test_tile_index.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "OK" in out.stdout
