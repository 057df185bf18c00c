"""Drop-in import paths of the reference (``from src.boundary import ...``,
``from src.chorin_fd.simulate import NavierStokesSystem`` ...) resolving to the MI355X-native
implementation in the sibling package ``nns``.  Put this directory's parent
(``neural-navier-stokes_amd/``) on PYTHONPATH in place of the reference checkout."""
