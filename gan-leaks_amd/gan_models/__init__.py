"""Generator classes with the reference's constructor / forward signatures (gan_models/*)."""
