/* oracle/refshim/cvsurf: shadows modules/xfeatures2d/include/opencv2/xfeatures2d.hpp (a dozen other feature classes over the main repo's
 * features2d.hpp); the SURF class declaration itself is the REFERENCE'S OWN header, included from where it lies. */
#include "features2d.hpp"
#include "opencv2/xfeatures2d/nonfree.hpp"
