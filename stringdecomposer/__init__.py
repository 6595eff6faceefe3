"""Import-name compatibility with the reference package: `stringdecomposer.main` is stringdecomposer_amd.main."""
