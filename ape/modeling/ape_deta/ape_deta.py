from ape_amd.modeling.ape_deta.ape_deta import SomeThing  # noqa: F401
