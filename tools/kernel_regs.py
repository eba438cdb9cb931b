"""VGPR / AGPR / scratch / LDS of the kernels in the built library whose mangled name contains a substring (the code-object metadata, read with the ROCm LLVM tools
like tests/test_host_logic.py does).   python tools/kernel_regs.py <substring> [library]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pat = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vmg_amd", "libvmg_hip.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "copy.so")], check=True)
    blob = open(fat, "rb").read()
    magic, starts, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    pos = blob.find(magic)
    while pos >= 0:
        starts.append(pos)
        pos = blob.find(magic, pos + 1)
    for i, st in enumerate(starts):
        end = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"b{i}.co")
        open(part, "wb").write(blob[st:end])
        r = subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"], capture_output=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines() + ["- .name: end"]:
            line = line.strip().lstrip("- ").strip()
            if line.startswith(".name:") or line.startswith("- .name:"):
                pass
            if ":" in line:
                k, v = line.split(":", 1)
                k = k.strip().lstrip("-").strip()
                if k == ".agpr_count" and cur.get(".name") and pat in cur.get(".name", ""):
                    pass
                if k in (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size", ".symbol"):
                    if k == ".symbol":
                        if pat in cur.get(".name", ""):
                            print(cur.get(".name", "?")[:110], {x: cur.get(x) for x in (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size")})
                        cur = {}
                    else:
                        cur[k] = v.strip()
