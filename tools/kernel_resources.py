#!/usr/bin/env python3
"""VGPRs / SGPRs / LDS / spills of the device kernels in a built library (the gfx950 code object inside the
fat binary): kernel_resources.py yalla_amd/libyalla_models.so [name filter]"""
import re
import struct
import subprocess
import sys
import tempfile

lib, pattern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
data = open(lib, "rb").read()
pos = data.index(b"__CLANG_OFFLOAD_BUNDLE__")
count = struct.unpack_from("<Q", data, pos + 24)[0]
off = pos + 32
for _ in range(count):
    o, size, tl = struct.unpack_from("<QQQ", data, off)
    off += 24
    triple = data[off:off + tl].decode()
    off += tl
    if "gfx950" in triple:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data[pos + o:pos + o + size])
            f.flush()
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True,
                                   text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, block).group(1)
            name = get("name")  # (c++filt does not know the mangling of non-type template arguments yet)
            if pattern in name:
                short = re.sub(r"15HIP_vector_typeIfLj(\d)EE", r"float\1", name)
                short = re.sub(r"TnPF[A-Za-z0-9_]*?EXadL_Z", " ", short).replace("_ZN2ya", "")
                print(f"vgpr {get('vgpr_count'):>3} spill {get('vgpr_spill_count')} sgpr {get('sgpr_count'):>3} "
                      f"lds {get('group_segment_fixed_size'):>6}  {short[:150]}")
