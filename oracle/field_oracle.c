/* TEST INFRASTRUCTURE ONLY -- placeholder, filled in with the BlobTree field / grid classification /
 * tetrahedral polygonizer restatement (see header of that section once present). */
int orc_field_placeholder(void) { return 0; }
