"""MI355X-native panoramic Gaussian-splat render path.  Importing the package loads nothing heavy; the HIP
library is loaded on first use (splatter360_amd._lib) and there is no CPU fallback."""


def install(**opts):
    """Register the fused decoder in the unchanged reference's decoder registry (splatter360_amd.plugin.install)."""
    from .plugin import install as _install
    return _install(**opts)


def uninstall():
    from .plugin import uninstall as _uninstall
    return _uninstall()
