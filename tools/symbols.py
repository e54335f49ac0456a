"""Short, stable names for this library's kernels as rocprofv3 reports them (mangled or demangled)."""
import re

_EPI = r"(EpiStore|EpiPartial|EpiPatch|EpiResidual|BigStore|BigPartial)"


def _prec(name):
    if "3hx2" in name or "hx2" in name:               # split-f16 operands (csrc/common.h): struct vitvs::hx2
        return "hx2"
    if "DF16b" in name or "__bf16" in name:
        return "bf16"
    if "DF16_" in name or "_Float16" in name:
        return "f16"
    return "f32"


def short(name):
    name = name.replace("vitvs::", "")
    m = re.match(r"_ZN5vitvs\d+(\w+?)I", name)
    plain = re.match(r"_ZN5vitvs(\d+)", name)
    if not m and plain:                                  # non-template kernel: _ZN5vitvs<len><name>E...
        n = int(plain.group(1))
        return name[len(plain.group(0)):len(plain.group(0)) + n]
    base = m.group(1) if m else re.sub(r"^void ", "", name).split("<")[0].split("(")[0]
    epi = re.search(_EPI, name)
    if base == "linear_big_kernel":
        t = re.search(r"BigTileILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"BigTile<(\d+), (\d+), (\d+), (\d+)>", name)
        dims = f"{int(t.group(1)) * int(t.group(3)) * 16}x{int(t.group(2)) * int(t.group(4)) * 16}" if t else "?"
        return f"linear_big_kernel<{_prec(name)},{dims}>" + (":" + epi.group(1) if epi else "")
    if base == "linear_kernel":
        dims = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r", (\d+), (\d+), (\d+),", name)
        return f"linear_kernel<{_prec(name)}" + ("," + ",".join(dims.groups()) if dims else "") + ">" + (":" + epi.group(1) if epi else "")
    if m or base.endswith("_kernel"):
        return f"{base}<{_prec(name)}>" if ("I" in name[len(base):] or "<" in name) else base
    return base[:60]
