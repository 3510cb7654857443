#!/usr/bin/env python3
"""Hand-patch the ISA of ONE kernel of quadrace_kernels.hip and link a library from it -- the experiment behind DESIGN's account of
the two-waves-per-SIMD corruption: insert wait states (or anything else) at a named place of a failing build WITHOUT letting the
compiler re-schedule or re-allocate anything else.

    python tools/isa_patch.py <source tree> <extra -D flags, comma separated or ''> <variant>=<spec>[;<spec>...] ...

A spec is  <kernel mangled-name regex>@<n>:<text>  = insert <text> (use '\\n' between instructions) BEHIND the n-th v_mfma of that
kernel (1-based), or  <regex>@<n>-:<text>  = in FRONT of it; an empty spec list builds the untouched ISA (the control).  Pipeline =
what `hipcc --save-temps` does: device .s -> cc1as -> lld -> clang-offload-bundler -> host .s (fat binary as .incbin) -> cc1as -> .so.
Libraries land in optimal_quad_control_rl_amd/_dbg/libisa_<variant>.so (they travel with gpurun; use QR_PROBE_LIB=... to load one)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fvisibility=hidden", "-fPIC"]
OTHERS = ["quadrace_abi.hip", "quadrace_policy.hip", "quadrace_ppo.hip", "quad3d.hip"]


def run(cmd, **kw):
    subprocess.check_call(cmd, **kw)


def main():
    tree, defs = os.path.abspath(sys.argv[1]), [d for d in sys.argv[2].split(",") if d]
    flags = list(FLAGS)
    if "NO_VGPR_FORM" in defs:
        defs.remove("NO_VGPR_FORM"); flags = [f for f in flags if f not in ("-mllvm", "-amdgpu-mfma-vgpr-form")]
    extra = []
    for d in defs:
        extra += (["-mllvm", d[5:]] if d.startswith("mllvm:") else ["-D" + d])
    tag = re.sub(r"[^A-Za-z0-9]+", "_", "_".join(sys.argv[2].split(","))) or "plain"
    out = os.path.join(tree, "_isa_" + tag); os.makedirs(out, exist_ok=True)
    csrc = os.path.join(tree, "optimal_quad_control_rl_amd", "csrc")
    objs = []
    for s in OTHERS:
        o = os.path.join(out, s.replace(".hip", ".o")); objs.append(o)
        if not os.path.exists(o):
            run(["/opt/rocm/bin/hipcc", *flags, *extra, "-c", os.path.join(csrc, s), "-o", o])
    dev_s = os.path.join(out, "quadrace_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
    host_s = os.path.join(out, "quadrace_kernels-host-x86_64-unknown-linux-gnu.s")
    if not os.path.exists(dev_s):
        run(["/opt/rocm/bin/hipcc", *flags, *extra, "--save-temps=obj", "-c", os.path.join(csrc, "quadrace_kernels.hip"), "-o", os.path.join(out, "k_orig.o")])
    base = open(dev_s).read().split("\n")
    host = open(host_s).read()
    for arg in sys.argv[3:]:
        name, _, specs = arg.partition("=")
        lines = list(base)
        for spec in [s for s in specs.split(";") if s]:
            if spec == "@fix":   # the product's rewrite of the hazardous packed-f32 forms (optimal_quad_control_rl_amd/isa_lint.py), nothing else
                sys.path.insert(0, ROOT)
                from optimal_quad_control_rl_amd import isa_lint
                text, nfix = isa_lint.fix_asm_text("\n".join(lines)); lines = text.split("\n")
                print(name, ":", nfix, "instructions rewritten")
                continue
            m = re.match(r"(.*?)@(re|sub):(.*)$", spec, re.S)
            if m:   # K@re:<regex>#<n>:<text> = insert behind the n-th line of the kernel matching regex;  K@sub:<regex>=><replacement>
                kre = m.group(1)
                start = next(i for i, l in enumerate(lines) if re.match(r"^(%s):" % kre, l))
                end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
                if m.group(2) == "sub":
                    rx, rep = m.group(3).split("=>")
                    hit = 0
                    for i in range(start, end):
                        new = re.sub(rx, rep, lines[i]); hit += new != lines[i]; lines[i] = new
                    assert hit, "no line matches " + rx
                else:
                    rx, rest = m.group(3).split("#", 1); n, text = rest.split(":", 1)
                    hits = [i for i in range(start, end) if re.search(rx, lines[i])]
                    assert len(hits) >= int(n), "only %d lines match %s" % (len(hits), rx)
                    lines.insert(hits[int(n) - 1] + 1, "\t" + text.replace("\\n", "\n\t"))
                continue
            m = re.match(r"(.*?)@(\d+)(-?):(.*)$", spec, re.S)
            kre, n, front, text = m.group(1), int(m.group(2)), m.group(3) == "-", m.group(4).replace("\\n", "\n\t")
            start = next(i for i, l in enumerate(lines) if re.match(r"^(%s):" % kre, l))
            end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
            mf = [i for i in range(start, end) if lines[i].lstrip().startswith("v_mfma")]
            at = mf[n - 1] if front else mf[n - 1] + 1
            lines.insert(at, "\t" + text)
        ps = os.path.join(out, "dev_%s.s" % name); open(ps, "w").write("\n".join(lines))
        po, pout, pfb = ps[:-2] + ".o", ps[:-2] + ".out", ps[:-2] + ".hipfb"
        run([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model", "pic", "-o", po, ps])
        run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", pout, po])
        run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=" + pout, "-output=" + pfb])
        # host side: swap the embedded fat binary (.asciz blob + .size) for an .incbin of the new one
        hs = re.sub(r'(\.L__unnamed_\d+):\n\t\.asciz\t"__CLANG_OFFLOAD_BUNDLE__.*?\n\t\.size\t\1, \d+',
                    lambda mm: '%s:\n\t.incbin\t"%s"\n\t.size\t%s, %d' % (mm.group(1), pfb, mm.group(1), os.path.getsize(pfb)), host, flags=re.S)
        assert hs != host, "fat binary blob not found in the host assembly"
        hps = os.path.join(out, "host_%s.s" % name); open(hps, "w").write(hs)
        ho = hps[:-2] + ".o"
        run([LLVM + "/clang", "-cc1as", "-triple", "x86_64-unknown-linux-gnu", "-filetype", "obj", "-target-cpu", "x86-64", "-mrelocation-model", "pic", "-o", ho, hps])
        lib = os.path.join(ROOT, "optimal_quad_control_rl_amd", "_dbg", "libisa_%s.so" % name)
        run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(csrc, "exports.map"), "-o", lib, ho, *objs])
        print("built", lib)


if __name__ == "__main__":
    main()
