"""Fortran namelist input as the reference reads it (`limits` / `ocn_in`, phy/mod_rdlim.F90:137-175 and the
&VCOORD, &DIFFUSION, &IDLGEO ... groups read by their modules): a small reader for the host side, and the mapping of
the groups' variables onto the library's options (include/blomgpu.h: blomgpu_set_real / _int / _str)."""
import re


def _value(tok):
    t = tok.strip()
    if not t:
        return None
    if t[0] in "'\"":
        return t[1:-1]
    u = t.lower()
    if u in (".true.", "t", ".t."):
        return True
    if u in (".false.", "f", ".f."):
        return False
    u = re.sub(r"_r?8$|_r8$", "", u).replace("d", "e")
    try:
        return int(u)
    except ValueError:
        return float(u)


def read_namelist(path):
    """{group: {variable: value | [values]}}, names in lower case; comments (!) and blank lines are skipped"""
    groups, cur = {}, None
    text = []
    for line in open(path):
        line = re.sub(r"!.*$", "", line) if "'" not in line else re.sub(r"(?<!')!(?=[^']*$).*$", "", line)
        text.append(line)
    body = "\n".join(text)
    for m in re.finditer(r"&(\w+)(.*?)^\s*/", body, re.S | re.M):
        cur = groups.setdefault(m.group(1).lower(), {})
        for a in re.finditer(r"(\w+)\s*=\s*((?:'[^']*'|[^=\n])+?)(?=\n\s*\w+\s*=|\s*$)", m.group(2).strip() + "\n", re.S):
            vals = [_value(v) for v in re.findall(r"'[^']*'|[^,\s]+", a.group(2))]
            vals = [v for v in vals if v is not None]
            cur[a.group(1).lower()] = vals[0] if len(vals) == 1 else vals
    return groups


# &LIMITS / &DIFFUSION / &VCOORD variables that are options of the dynamical core
REALS = ("pref", "baclin", "batrop", "mdv2hi", "mdv2lo", "mdv4hi", "mdv4lo", "mdc2hi", "mdc2lo", "vsc2hi", "vsc2lo",
         "vsc4hi", "vsc4lo", "cbar", "cb", "cwbdts", "cwbdls", "bdmc1", "bdmc2", "iwdfac", "nubmin")
STRS = ("expcnf", "mommth", "pgfmth", "bmcmth", "advmth", "cppm_compatibility", "cppm_limiting", "eitmth")
INTS = ("bdmtyp", "iwdflg")


def options_from_namelists(groups):
    """the library options a `limits` file sets (keys as blomgpu_set_* takes them)"""
    lim, dif, vc = groups.get("limits", {}), groups.get("diffusion", {}), groups.get("vcoord", {})
    out = {}
    for src in (lim, dif):
        for k in REALS:
            if k in src:
                out[k] = float(src[k])
        for k in STRS:
            if k in src:
                out[k] = str(src[k])
        for k in INTS:
            if k in src:
                out[k] = int(src[k])
    if "bdmldp" in dif:
        out["bdmldp"] = 1 if dif["bdmldp"] else 0
    if "ltedtp" in dif:
        out["ltedtp_opt"] = {"layer": 1, "neutral": 2}[dif["ltedtp"]]
    if "cnsvdi" in lim:
        out["cnsvdi"] = 1 if lim["cnsvdi"] else 0
    if "vcoord_type" in vc:
        out["vcoord_type"] = vc["vcoord_type"]
    return out
