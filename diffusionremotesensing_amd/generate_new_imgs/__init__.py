"""Mirror of the reference's generate_new_imgs/ package (class-conditional image generation)."""
