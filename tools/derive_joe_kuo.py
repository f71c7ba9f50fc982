#!/usr/bin/env python3
"""The Joe-Kuo direction numbers (S. Joe, F. Y. Kuo 2008, table new-joe-kuo-6: degree s, coefficient bits a, initial numbers
m_1 .. m_s per dimension) of the Sobol' dimensions 2 .. N, RECOVERED from the generator matrices the reference ships
(/root/reference/src/core/sobolmatrices.rs:81, SOBOL_MATRICES32: 52 columns per dimension, column i = m_i << (32 - i)): the first s
columns of a dimension are its initial numbers, and (s, a) is the one primitive polynomial whose recurrence
    m_i = XOR_{k=1..s-1} a_k 2^k m_{i-k}  ^  2^s m_{i-s}  ^  m_{i-s}
reproduces columns s + 1 .. 32.  Prints the rows in the form pbrt_amd/csrc/host_math.hpp and oracle/oracle.cpp hold them (both
build their matrices from these ~10 small integers per dimension; neither holds the 53 248-word table), for N dimensions.
Run where the reference is mounted:   python tools/derive_joe_kuo.py 128
tests/test_host.py::test_sobol_nd_generator_matrices and tests/test_reference_vectors.py compare the matrices built from the rows
with the reference's table entry for entry."""
import re
import sys

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
src = open("/root/reference/src/core/sobolmatrices.rs").read()
body = src[src.index("const SOBOL_MATRICES32"):src.index("const SOBOL_MATRICES64")]
words = [int(w, 16) for w in re.findall(r"0x([0-9a-fA-F]+)", body[body.index("=") :])]
assert len(words) == 1024 * 52, len(words)
rows = []
for d in range(1, N):
    col = words[d * 52:d * 52 + 52]
    m = [None] + [col[i - 1] >> (32 - i) for i in range(1, 33)]
    found = None
    for s in range(1, 14):
        for a in range(0, 1 << max(s - 1, 0)):
            ok = True
            for i in range(s + 1, 33):
                v = m[i - s] ^ (m[i - s] << s)
                for k in range(1, s):
                    if (a >> (s - 1 - k)) & 1:
                        v ^= m[i - k] << k
                if v != m[i]:
                    ok = False
                    break
            if ok:
                found = (s, a)
                break
        if found:
            break
    assert found, d
    s, a = found
    rows.append((s, a, m[1:s + 1]))
line = []
for s, a, mm in rows:
    line.append("{%d, %d, {%s}}" % (s, a, ", ".join(map(str, mm))))
for i in range(0, len(line), 3):
    print("    " + ", ".join(line[i:i + 3]) + ",")
print("// max degree", max(r[0] for r in rows), file=sys.stderr)
