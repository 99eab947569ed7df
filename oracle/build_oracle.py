"""Builds oracle/oph_cpu.c (the C restatement used as CPU baseline / second checker)
into oracle/_build/liboph_cpu.so with gcc + OpenMP.  Test infrastructure only."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboph_cpu.so")
STAMP = os.path.join(OUT, "stamp.txt")


def _cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return line
    except OSError:
        pass
    return ""


def build(native=True):
    """native=True -> -march=native for THIS host (rebuilt when the host CPU differs)."""
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "oph_cpu.c")
    march = "native" if native else "x86-64-v3"
    key = hashlib.sha1((open(src).read() + march + (_cpu_flags() if native else "")).encode()).hexdigest()
    if os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == key:
        return LIB
    cmd = ["gcc", "-O3", "-march=" + march, "-fopenmp", "-fPIC", "-shared", "-std=c99", src, "-lm", "-o", LIB]
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(key)
    return LIB


if __name__ == "__main__":
    print(build())
