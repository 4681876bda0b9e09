"""Readable names for this library's kernels in rocprofv3 outputs."""
import re


def demangle(n):
    """Enough of the Itanium ABI for this library's kernels (binutils' c++filt does not know DF16_/DF16b)."""
    m = re.match(r"_ZN4fvit12_GLOBAL__N_1(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    start = m.end()
    name, rest = n[start:start + ln], n[start + ln:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while rest and not rest.startswith("E"):
            for pat, fn in ((r"DF16_", lambda g: "f16"), (r"DF16b", lambda g: "bf16"), (r"Li(\d+)E", lambda g: g.group(1)),
                            (r"Lb([01])E", lambda g: "true" if g.group(1) == "1" else "false")):
                mm = re.match(pat, rest)
                if mm:
                    args.append(fn(mm))
                    rest = rest[mm.end():]
                    break
            else:
                break
    return name + ("<" + ",".join(args) + ">" if args else "")
