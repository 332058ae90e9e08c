"""Drop-in contract at the LazyConfig level: every keyword the reference's APE-L_D / APE-Ti configuration chain passes
to a hot-path class -- `L(Class)(kw=...)` calls and later `model.model_vision[.transformer[.encoder|.decoder]].<kw> = ...`
assignments -- is accepted by the HIP-backed class of the same name.  Static (AST) check of the reference's config files; needs
the reference checkout."""
import ast
import inspect
import os
import re

import pytest

REF = os.environ.get("APE_REFERENCE", "/root/reference")
CFG = os.path.join(REF, "configs")
CHAINS = {
    "L_D": ["COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py",
            "LVIS_InstanceSegmentation/ape_deta/ape_deta_vitl_eva02_lsj1024_cp_24ep.py",
            "LVISCOCOCOCOSTUFF_O365_OID_VGR_SA1B_REFCOCO_GQA_PhraseCut_Flickr30k/ape_deta/ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py",
            "common/backbone/vitl_eva02_clip.py"],
    "Ti": ["COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py",
           "LVISCOCOCOCOSTUFF_O365_OID_VGR_SA1B_REFCOCO_GQA_PhraseCut_Flickr30k/ape_deta/ape_deta_vitt_eva02_vlf_lsj1024_cp_16x4_1080k.py",
           "common/backbone/vitt_eva02.py"],
    "L_A": ["common/backbone/vitl_eva02.py"],
    # f4b / f4c: the ViT-g / ViT-e backbone configurations (EVA-01-CLIP ViT-g, ViT-e: vit_eva_clip classes; EVA-01 MIM ViT-g: vit_eva.py)
    "G_A": ["common/backbone/vitg_eva01_clip_1024.py", "common/backbone/vitg_eva01_clip_1536.py", "common/backbone/vite_eva02_clip_1024.py"],
    "V_A": ["common/backbone/vitg_eva01.py", "common/backbone/vitg_eva01_1536.py"],
}
# config class name -> (our class, attribute path whose later assignments also count)
pytestmark = pytest.mark.skipif(not os.path.isdir(CFG), reason="needs the reference checkout")


def _collect(files):
    calls, assigns = {}, {}
    for rel in files:
        path = os.path.join(CFG, rel)
        if not os.path.isfile(path):
            continue
        src = open(path).read()
        for node in ast.walk(ast.parse(src)):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Call) and getattr(node.func.func, "id", "") == "L" and node.func.args:
                tgt = node.func.args[0]
                name = getattr(tgt, "id", None) or getattr(tgt, "attr", None)
                if name:
                    calls.setdefault(name, set()).update(k.arg for k in node.keywords if k.arg)
        for m in re.finditer(r"^model\.model_vision((?:\.\w+)*)\.(\w+)\s*=", src, flags=re.M):
            assigns.setdefault(m.group(1), set()).add(m.group(2))
    return calls, assigns


def _accepts(cls):
    sig = inspect.signature(cls.__init__)
    if any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()):
        return None
    return set(sig.parameters) - {"self"}


@pytest.mark.parametrize("chain", ["L_D", "Ti", "L_A", "G_A", "V_A"])
def test_every_config_keyword_is_accepted(chain):
    from ape_amd.layers import VisionLanguageFusion
    from ape_amd.modeling.ape_deta import (DeformableDETRSegmVL, DeformableDetrTransformerDecoderVL, DeformableDetrTransformerEncoderVL,
                                           DeformableDetrTransformerVL)
    from ape_amd.modeling.backbone import vit_eva, vit_eva02, vit_eva_clip
    from ape_amd.modeling.text import EVA02CLIP

    calls, assigns = _collect(CHAINS[chain])
    vit_mod = vit_eva_clip if chain in ("L_D", "G_A") else vit_eva if chain == "V_A" else vit_eva02
    table = [
        (("DeformableDETRSegm", "DeformableDETRSegmVL"), "", DeformableDETRSegmVL),
        (("DeformableDetrTransformer", "DeformableDetrTransformerVL"), ".transformer", DeformableDetrTransformerVL),
        (("DeformableDetrTransformerEncoder", "DeformableDetrTransformerEncoderVL"), ".transformer.encoder", DeformableDetrTransformerEncoderVL),
        (("DeformableDetrTransformerDecoder", "DeformableDetrTransformerDecoderVL"), ".transformer.decoder", DeformableDetrTransformerDecoderVL),
        (("VisionLanguageFusion",), None, VisionLanguageFusion),
        (("ViT",), None, vit_mod.ViT),
        (("SimpleFeaturePyramid",), None, vit_mod.SimpleFeaturePyramid),
        (("EVA02CLIP",), None, EVA02CLIP),
    ]
    checked = 0
    for names, path, cls in table:
        want = set()
        for n in names:
            want |= calls.get(n, set())
        if path is not None:
            want |= assigns.get(path, set())
        want -= {"_target_"}
        # sub-configs replaced wholesale by later configs are not keywords of the class itself
        if cls is DeformableDETRSegmVL:
            want -= {"transformer", "backbone", "neck"} - set(inspect.signature(cls.__init__).parameters)
        ok = _accepts(cls)
        if ok is None or not want:
            continue
        missing = sorted(want - ok)
        assert not missing, f"{chain}: {cls.__name__} does not accept {missing}"
        checked += len(want)
    assert checked >= (8 if chain in ("L_A", "G_A", "V_A") else 40), checked      # backbone-only chains: the ViT + pyramid keywords


def test_overrides_of_every_vit_ape_config_are_accepted():
    """all configs/**/ape_deta/*vit[lt]_eva02*.py (APE-Ti, APE-L_A..D, every dataset mix): each `model.model_vision[.transformer
    [.encoder|.decoder]].<kw> = ...` override names a constructor keyword of the HIP-backed class"""
    import glob
    from ape_amd.modeling.ape_deta import (DeformableDETRSegmVL, DeformableDetrTransformerDecoderVL, DeformableDetrTransformerEncoderVL,
                                           DeformableDetrTransformerVL)
    acc = {"": DeformableDETRSegmVL, ".transformer": DeformableDetrTransformerVL, ".transformer.encoder": DeformableDetrTransformerEncoderVL,
           ".transformer.decoder": DeformableDetrTransformerDecoderVL}
    ok = {k: set(inspect.signature(c.__init__).parameters) for k, c in acc.items()}
    files = glob.glob(os.path.join(CFG, "**", "ape_deta", "*vit[lt]_eva02*.py"), recursive=True)
    assert len(files) > 50
    bad, n = {}, 0
    for f in files:
        for m in re.finditer(r"^model\.model_vision((?:\.\w+)*)\.(\w+)\s*=", open(f).read(), flags=re.M):
            if m.group(1) in ok:
                n += 1
                if m.group(2) not in ok[m.group(1)]:
                    bad.setdefault((m.group(1), m.group(2)), []).append(os.path.basename(f))
    assert not bad and n > 500, (bad, n)
