typedef unsigned char GLubyte; namespace pangolin { struct GlTexture; }
