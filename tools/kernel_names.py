"""rocprofv3 kernel name -> the name libsgan_hip reports through sgan_last_kernel() (shared by bench.py and the profile tools)."""
import re


def short(name):
    """rocprofv3 kernel name -> the name libsgan_hip reports through sgan_last_kernel() (template variants of one kernel
    merged: the prologue flag of sg_igemm / sg_wgrad, the layout flag of sg_conv_small_n)."""
    n = name.split("(")[0].replace("void ", "").replace(" ", "")
    m = re.match(r"(sg_igemm_kernel)<(\d+,\d+,\d+,\d+,(?:true|false)),(?:true|false)(?:,\d+)?>$", n)   # prologue flag, wave groups
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"(sg_igemm3_kernel)<(\d+,\d+,\d+,\d+),(?:true|false),(?:true|false)>$", n)   # prologue flag, fp16 / bf16 planes
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"(sg_igemm3p)(?:_kw2)?_kernel<(\d+),(.*)>$", n)      # N tile, staging passes, prologue flag, plane type, stride-2 flag
    if m:
        return f"{m.group(1)}_kernel<{m.group(2)},s2>" if m.group(3).endswith(",true") and m.group(3).count(",") == 3 else f"{m.group(1)}_kernel<{m.group(2)}>"
    m = re.match(r"(sg_igemm3p_kernel)<(\d+),.*>$", n)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"(sg_wgrad3_kernel)<(\d+,\d+,\d+,\d+),(?:true|false)>$", n)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"sg_bwd_fused_kernel<(\d+),\d+,(?:true|false)(?:,(?:true|false))?>$", n)      # backward-data variant, backward-weight tile, prologue flag, fp16 planes
    if m:
        return "sg_bwd_fused_kernel<f32 dgrad>" if m.group(1) == "4" else "sg_bwd_fused_kernel"
    m = re.match(r"(sg_wgrad_kernel)<(.*),(true|false)>$", n)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"sg_conv_c4_kernel<\d+,\d+,(?:true|false)>$", n)      # column blocks, row blocks per wave, full epilogue
    if m:
        return "sg_conv_c4_kernel"
    if re.match(r"sg_bwd_thin_pair_kernel<.*>$", n):      # backward-data body, row blocks, thin operand width, prologue flag, operand swap
        return "sg_bwd_thin_pair_kernel"
    if re.match(r"sg_head_bwd_kernel<\d+>$", n):      # kernel size
        return "sg_head_bwd_kernel"
    m = re.match(r"sg_conv_head_kernel<\d+>$", n)      # tile height
    if m:
        return "sg_conv_head_kernel"
    m = re.match(r"sg_conv_small_n_kernel<(\d+),(?:\d+,)*(true|false)>$", n)
    if m:
        return f"sg_conv_small_n_kernel<{m.group(1)}>"
    m = re.match(r"sg_wgrad_thin_kernel<(\d+),(true|false),(true|false)>$", n)
    if m:
        return f"sg_wgrad_thin_kernel<{m.group(1)},{'cout4' if m.group(3) == 'true' else 'cin4'}>"
    return n
